import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mujoco_robot_environments_amd import rng
from mujoco_robot_environments_amd.model import compile as MC
from oracle import oracle as O
from tests.common import init_oracle_env
O.build()
om = O.Model(MC.to_blob(MC.compile_scene()))
scale, seed = 1.0, 5
N, T, cs = 256, 200, 5
ids = np.arange(N)
nprops, sizes = rng.prop_params(seed, ids)
acts = rng.random_actions(seed, ids, np.arange(T), scale=scale)
yaws = rng.uniform(seed + 7, ids, [0], 4)[0] * np.pi
a32 = acts.astype(np.float32).astype(np.float64)
def run(i, solver, eps, at=400):
    e = O.Env(om, int(nprops[i]), sizes[i]); e.set_solver(solver)
    q0 = init_oracle_env(e, int(nprops[i]), sizes[i], z_extra=0.0005, yaw=yaws[i]).astype(np.float32)
    e.arr("qpos")[:43] = q0; e.forward()
    out = np.zeros((T*cs, 43))
    for t in range(T):
        e.arr("ctrl")[:] = a32[t, i]
        for k in range(cs):
            if t*cs+k == at: e.arr("qvel")[7:15] += eps * np.array([1,-1,1,-1,1,-1,1,-1.])
            e.step(1); out[t*cs+k] = e.arr("qpos")[:43]
    return out

import concurrent.futures as cf
def amp(i):
    a = run(i, "PGS", 0.0); b = run(i, "PGS", 1e-9)
    d = np.abs(a-b)[:, :15].max(axis=1)
    return i, d[450], d[700], d[999], d[400:].max()
envs = [34,142,119,193] + list(range(0,64))
with cf.ThreadPoolExecutor(8) as ex:
    res = list(ex.map(amp, envs))
for r in res[:4]: print("env %d: |dq| robot at 450 %.1e, 700 %.1e, 999 %.1e, max %.1e" % r)
m = np.array([r[4] for r in res[4:]])
print("envs 0..63: max |dq| quantiles 50/75/90/97/100 %%: %s" % " ".join(f"{np.quantile(m,q):.1e}" for q in (0.5,0.75,0.9,0.97,1.0)))
print("envs with max > 3e-9:", [(r[0], f"{r[4]:.1e}") for r in res[4:] if r[4] > 3e-9])
