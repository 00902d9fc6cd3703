import sys, numpy as np
sys.path.insert(0, '.')
from tests import test_gpu_parity as T
from mujoco_robot_environments_amd.model import compile as MC
from oracle import oracle as O
A = MC.compile_scene(); blob = MC.to_blob(A)
om = O.Model(blob)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 40
gq, oq, nprops, phys = T._rollout_both((A, blob), om, N=N, T=Tn, flags=0, scale=0.1, z_extra=0.002,
                                       gravity_comp=True, yaw=True)
err = np.abs(gq - oq)
for i in range(N):
    err[:, i, 15 + 7 * int(nprops[i]):] = 0
print("status", phys.status()[:N], "stats", phys.solver_stats()[:N].tolist())
for i in range(N):
    e = err[:, i]
    k = np.unravel_index(e.argmax(), e.shape)
    print(f"env {i} nprops {nprops[i]} max {e.max():.3e} at step {k[0]} q{k[1]} | arm {e[:, :7].max():.2e} grip {e[:, 7:15].max():.2e} cubes {e[:, 15:].max():.2e}")
i = int(np.argmax(err.max(axis=(0, 2))))
print("worst env", i)
for t in [0, 1, 2, 5, 10, 20, 50, 100, 199]:
    if t < err.shape[0]:
        print(t, "arm %.2e grip %.2e cubepos %.2e" % (err[t, i, :7].max(), err[t, i, 7:15].max(), err[t, i, 15:].max()),
              "gpu z", gq[t, i, 17], "orc z", oq[t, i, 17])
if len(sys.argv) > 3:
    i = int(sys.argv[3])
    np.set_printoptions(precision=6, suppress=False, linewidth=200)
    e = err[:, i, 7:15]
    first = np.argmax(e.max(axis=1) > 2e-5)
    print("first step with grip err > 2e-5:", first)
    for t in range(max(0, first - 6), min(err.shape[0], first + 8)):
        print(t, "err", e[t], "\n    orc", oq[t, i, 7:15], "\n    gpu", gq[t, i, 7:15])
