import sys, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.physics import BatchedPhysics
from mujoco_robot_environments_amd.model import compile as MC
A = MC.compile_scene()
phys = BatchedPhysics(2, model=A)
rows = np.zeros((2, 4, 7))
for p in range(4):
    rows[0, p] = [0.45 + 0.029 * p, 0.0, 0.4135, 1, 0, 0, 0]
    rows[1, p] = [0.40 + 0.05 * p, -0.3 + 0.2 * p, 0.4155, 1, 0, 0, 0]
qp = phys.qpos().copy()
for i in range(2):
    qp[i, :7] = A['home_qpos']; qp[i, 15:43] = rows[i].reshape(-1)
phys.set_state(qp, np.zeros((2, 39), np.float32))
phys.step(1)
print("status", phys.status(), "stats", phys.solver_stats().tolist(), phys.fallback_stats())
phys.step(1)
print("status", phys.status(), "stats", phys.solver_stats().tolist(), phys.fallback_stats())
