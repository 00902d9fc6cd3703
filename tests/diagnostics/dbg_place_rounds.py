import sys, numpy as np
sys.path.insert(0, '.')
import bench
from mujoco_robot_environments_amd.physics import BatchedPhysics
for solver in ("PGS", "Newton"):
    print("==", solver, flush=True)
    phys = BatchedPhysics(4096, solver=solver)
    bench.setup_envs(phys, 0, np.arange(4096))
    st = phys.status()
    print("status bits:", {int(b): int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16)}, flush=True)
    phys.close()
