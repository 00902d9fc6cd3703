"""Is the Newton bench bounded by ONE env's serial latency?  Every env's own duration per tick (mre_get_launch_info)
summed over the default bench's ticks: max_e sum_t d_e(t) is the floor of any schedule that keeps an env's ticks in
order; sum_t max_{e in group} d_e(t) is what a launch-per-tick pipeline of fixed groups pays.
usage: python tests/diagnostics/dbg_env_serial_time.py [PGS|Newton] [nticks=220] [warmup=20]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.physics import BatchedPhysics

solver = sys.argv[1] if len(sys.argv) > 1 else "Newton"
nticks = int(sys.argv[2]) if len(sys.argv) > 2 else 220
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N = 4096
phys = BatchedPhysics(N, solver=solver); ids = np.arange(N)
bench.setup_envs(phys, 0, ids)
seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(nticks)).astype(np.float32)).to(phys.device).contiguous()
D = np.zeros((nticks, N))
for t in range(nticks):
    phys.rollout(seq[t:t + 1], control_steps=5)
    D[t] = phys.launch_info()["duration"].astype(np.float64)
D = D[warm:]
unit = 16 / 2.4e9 * 1e3   # ms per unit of 16 cycles at 2.4 GHz
tot = D.sum(axis=0)
print(f"{solver}: {D.shape[0]} ticks; mean env-tick {D.mean() * unit:.3f} ms; throughput floor sum/2048 slots = {D.sum() / 2048 / D.shape[0] * unit:.3f} ms/tick")
print(f"  slowest env's serial time: {tot.max() / D.shape[0] * unit:.3f} ms/tick (env {tot.argmax()}), p99 {np.percentile(tot, 99) / D.shape[0] * unit:.3f}, median {np.median(tot) / D.shape[0] * unit:.3f}")
print(f"  sum over ticks of the batch's slowest env: {D.max(axis=1).sum() / D.shape[0] * unit:.3f} ms/tick")
for G in (4, 16):
    g = D.reshape(D.shape[0], G, N // G).max(axis=2).sum(axis=0)
    print(f"  {G} groups of consecutive envs: per-group chain max {g.max() / D.shape[0] * unit:.3f} mean {g.mean() / D.shape[0] * unit:.3f} ms/tick")
top = np.argsort(-tot)[:5]
print("  top envs (ms/tick):", [(int(i), round(tot[i] / D.shape[0] * unit, 3)) for i in top])
print("  how often is the batch's slowest env one of the 5 slowest overall:", np.isin(D.argmax(axis=1), top).mean())
