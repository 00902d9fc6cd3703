import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.physics import BatchedPhysics
N = 4096
phys = BatchedPhysics(N)
ids = np.arange(N)
nprops, _ = bench.setup_envs(phys, 0, ids)
T = 220
acts = rng.random_actions(0, ids, np.arange(T)).astype(np.float32)
seq = torch.from_numpy(acts).to(phys.device).contiguous()
for t0 in (0, 20, 60, 120, 200):
    while phys_t[0] < t0 if (phys_t := globals().setdefault('phys_t', [0])) else False:
        phys.rollout(seq[phys_t[0]:phys_t[0] + 1], control_steps=5); phys_t[0] += 1
    phys.sync()
    phys.profile_enable(True)
    for k in range(5):
        phys.rollout(seq[phys_t[0]:phys_t[0] + 1], control_steps=5); phys_t[0] += 1
    ms, n = phys.profile_read(); phys.profile_enable(False)
    st = phys.solver_stats()
    nefc, ncon, nl = st[:, 1], st[:, 0], st[:, 3]
    print(f"tick {t0}: {ms / n:.2f} ms/launch | nefc mean {nefc.mean():.1f} p90 {np.percentile(nefc, 90):.0f} p99 {np.percentile(nefc, 99):.0f} max {nefc.max()} | ncon max {ncon.max()} | nl mean {nl.mean():.2f} max {nl.max()} | overflow {(phys.status() & 4 != 0).sum()}")
