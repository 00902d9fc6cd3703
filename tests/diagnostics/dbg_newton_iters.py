"""Newton iterations per step, device vs oracle, on the bench's action law (last step of every tick).
usage: python tests/diagnostics/dbg_newton_iters.py [nenvs] [nticks]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.model import compile as MC
from mujoco_robot_environments_amd.physics import BatchedPhysics
from oracle import oracle as O
from tests.common import init_oracle_env

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed = 5
A = MC.compile_scene()
om = O.Model(MC.to_blob(A))
ids = np.arange(N)
nprops, sizes = rng.prop_params(seed, ids)
acts = rng.random_actions(seed, ids, np.arange(T), scale=1.0).astype(np.float32)
yaws = rng.uniform(seed + 7, ids, [0], 4)[0] * np.pi
envs, q0 = [], np.zeros((N, 43))
for i in range(N):
    e = O.Env(om, int(nprops[i]), sizes[i]); e.set_solver("Newton")
    q0[i] = init_oracle_env(e, int(nprops[i]), sizes[i], z_extra=0.0005, yaw=yaws[i])
    envs.append(e)
phys = BatchedPhysics(N, model=A, solver="Newton")
phys.set_props(nprops, sizes)
qp = phys.qpos().copy()
for i in range(N):
    n = int(nprops[i]); qp[i, :15 + 7 * n] = q0[i, :15 + 7 * n]
    envs[i].arr("qpos")[:43] = qp[i]; envs[i].forward()
phys.set_state(qp, np.zeros((N, 39), np.float32))
seq = torch.from_numpy(acts).to(phys.device).contiguous()
gi, oi, gf = [], [], []
for t in range(T):
    phys.rollout(seq[t:t + 1], control_steps=5)
    st = phys.solver_stats()
    gi.append(st[:, 2].copy()); gf.append(phys.last_factorizations.copy())
    row = []
    for i, e in enumerate(envs):
        e.arr("ctrl")[:] = acts[t, i].astype(np.float64)
        e.step(5)
        row.append(e.solver_iters)
    oi.append(row)
gi, oi, gf = np.array(gi), np.array(oi), np.array(gf)
print(f"iterations per step (last step of each tick), {N} envs x {T} ticks: device mean {gi.mean():.2f} max {gi.max()}, oracle mean {oi.mean():.2f} max {oi.max()}; "
      f"device factorisations mean {gf.mean():.2f}")
for lo in (1, 3, 5, 8):
    m = oi >= lo
    print(f"  oracle iters >= {lo}: {m.mean() * 100:.1f} % of steps; device iters there: mean {gi[m].mean():.2f} vs oracle {oi[m].mean():.2f}")
d = gi.astype(int) - oi.astype(int)
print("  device - oracle iterations: histogram", {int(k): int((d == k).sum()) for k in np.unique(d)})
