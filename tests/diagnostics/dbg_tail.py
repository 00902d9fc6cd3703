"""Per-tick launch time vs. the heaviest envs of the batch (lock-step tail of the random-action workload)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.physics import BatchedPhysics
N = 4096
phys = BatchedPhysics(N); ids = np.arange(N)
bench.setup_envs(phys, 0, ids)
T = 220
seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(T)).astype(np.float32)).to(phys.device).contiguous()
for t0 in range(0, T, 20):
    phys.profile_enable(True)
    for t in range(t0, t0 + 20):
        phys.rollout(seq[t:t + 1], control_steps=5)
    ms, n = phys.profile_read()
    st = phys.solver_stats()
    q = phys.qpos()
    nefc = st[:, 1]
    print(f"ticks {t0:3d}-{t0+19:3d}: {ms/n:6.2f} ms/tick | nefc mean {nefc.mean():5.1f} p99 {np.percentile(nefc,99):5.0f} max {nefc.max():3d} | envs nefc>70: {(nefc>70).sum():3d} >90: {(nefc>90).sum():3d} | {phys.fallback_stats()}")
