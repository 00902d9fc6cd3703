import sys, numpy as np
sys.path.insert(0, '.')
from tests import test_gpu_parity as T
from mujoco_robot_environments_amd.model import compile as MC
from oracle import oracle as O
A = MC.compile_scene(); blob = MC.to_blob(A)
om = O.Model(blob)
gq, oq, nprops, phys = T._rollout_both((A, blob), om, N=64, T=200, flags=0, scale=0.1, seed=21, z_extra=0.0005,
                                       gravity_comp=True, yaw=True)
err = np.abs(gq - oq)[:, :, 7:15]
rng = A['jnt_range'][8:16]
names = A['_names']['bodies'][8:16]
np.set_printoptions(precision=6, suppress=True, linewidth=220)
bad = [i for i in range(64) if err[:, i].max() > 1e-4]
print("bad envs", len(bad))
for i in bad[:12]:
    t0 = int(np.argmax(err[:, i].max(axis=1) > 2e-5))
    j = int(np.argmax(err[t0, i]))
    # which joint is nearest to a limit (in the oracle) in the 3 steps before t0
    q = oq[max(t0 - 2, 0):t0 + 1, i, 7:15]
    dlo, dhi = q - rng[:, 0], rng[:, 1] - q
    near = np.minimum(np.abs(dlo), np.abs(dhi)).min(axis=0)
    k = int(np.argmin(near))
    print(f"env {i}: first>2e-5 at step {t0} joint {names[j]} err {err[t0, i, j]:.2e}; nearest-to-limit joint {names[k]} |dist| {near[k]:.2e} "
          f"(oracle {min(dlo[-1, k], dhi[-1, k]):+.2e}, gpu {min(gq[t0, i, 7 + k] - rng[k, 0], rng[k, 1] - gq[t0, i, 7 + k]):+.2e}); final err {err[-1, i].max():.2e} max {err[:, i].max():.2e}")
