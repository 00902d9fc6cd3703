"""Which envs end a launch?  Per-env durations of late bench ticks (mre_get_launch_info) next to their
constraint counts and solver iterations.  usage: python tests/diagnostics/dbg_stragglers.py [PGS|Newton] [nticks]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.physics import BatchedPhysics

solver = sys.argv[1] if len(sys.argv) > 1 else "Newton"
nticks = int(sys.argv[2]) if len(sys.argv) > 2 else 220
N = 4096
phys = BatchedPhysics(N, solver=solver); ids = np.arange(N)
bench.setup_envs(phys, 0, ids)
seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(nticks)).astype(np.float32)).to(phys.device).contiguous()
for t in range(nticks):
    phys.rollout(seq[t:t + 1], control_steps=5)
    if t in (20, 100, 200, nticks - 1):
        li = phys.launch_info(); st = phys.solver_stats()
        d = li["duration"].astype(np.float64)
        order = np.argsort(-d)
        print(f"tick {t}: duration mean {d.mean():.0f} median {np.median(d):.0f} p90 {np.percentile(d, 90):.0f} p99 {np.percentile(d, 99):.0f} "
              f"max {d.max():.0f}; sum/2048 slots = {d.sum() / 2048:.0f} (two balanced rounds) vs max {d.max():.0f}")
        for i in order[:8]:
            print(f"   env {i}: duration {d[i]:.0f} ({d[i] / d.mean():.1f}x mean) ncon {li['ncon'][i]} nefc {li['nefc'][i]} nrrow {li['nrrow'][i]} "
                  f"npp {li['npp'][i]} last-step iters {st[i, 2]} factorisations {phys.last_factorizations[i] if solver == 'Newton' else '-'} large {phys.fallback_stats()}")
