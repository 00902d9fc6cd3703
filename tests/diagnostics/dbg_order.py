import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.physics import BatchedPhysics
N = 4096
res = {}
for mode in ("identity", "heavy_first"):
    phys = BatchedPhysics(N)
    ids = np.arange(N)
    bench.setup_envs(phys, 0, ids)
    T = 220
    seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(T)).astype(np.float32)).to(phys.device).contiguous()
    phys.profile_enable(True)
    times = []
    for t in range(T):
        if mode == "heavy_first" and t % 5 == 0 and t > 0:
            st = phys.solver_stats()
            phys.set_env_order(np.argsort(-st[:, 1], kind="stable"))
        phys.rollout(seq[t:t + 1], control_steps=5)
        if t % 20 == 19:
            ms, n = phys.profile_read(); times.append(ms / n)
    res[mode] = times
    print(mode, [round(x, 2) for x in times], "mean", round(float(np.mean(times)), 2))
    phys.close()
