"""Long-run stability of the step kernels on the bench workload: N ticks of full-range random actions,
status / finiteness / capacity checked along the way.  usage: python tools/soak.py [PGS|Newton] [nticks]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.physics import BatchedPhysics

solver = sys.argv[1] if len(sys.argv) > 1 else "Newton"
nticks = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
N, CH = 4096, 500
phys = BatchedPhysics(N, solver=solver); ids = np.arange(N)
bench.setup_envs(phys, 0, ids)
t0 = time.time()
for c0 in range(0, nticks, CH):
    seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(c0, c0 + CH)).astype(np.float32)).to(phys.device).contiguous()
    for t in range(CH):
        phys.rollout(seq[t:t + 1], control_steps=5)
    phys.sync()
    st = phys.status(); qp = phys.qpos(); ss = phys.solver_stats()
    z = qp[:, 15:43].reshape(N, 4, 7)[:, :, 2]
    nprops = rng.prop_params(0, ids)[0]
    inuse = np.arange(4)[None, :] < nprops[:, None]        # (the unused cube slots are parked far below the floor)
    print(f"tick {c0 + CH}: {time.time() - t0:.1f} s, nan {int(((st & 2) != 0).sum())}, overflow {int(((st & 4) != 0).sum())}, finite {bool(np.isfinite(qp).all())}, "
          f"mean ncon {ss[:, 0].mean():.1f} max {ss[:, 0].max()}, max nefc {ss[:, 1].max()}, iters mean {ss[:, 2].mean():.2f} max {ss[:, 2].max()}, "
          f"cubes in use below the table top {int(((z < 0.39) & inuse).sum())}, |qvel| max {np.abs(phys.qvel()).max():.1f}, fallback {phys.fallback_stats()}", flush=True)
