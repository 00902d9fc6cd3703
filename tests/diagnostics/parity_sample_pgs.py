"""The two 1000-step parity runs of tests/test_gpu_newton.py with the PGS solver (north_star's solver, cut at 100
sweeps), same seeds and report.  usage: python tests/diagnostics/parity_sample_pgs.py [nenvs=64]   (GPU; not a test)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mujoco_robot_environments_amd.model import compile as MC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.test_gpu_newton import TOL, _divergence_report  # noqa: E402
from tests.test_gpu_parity import _rollout_both  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A = MC.compile_scene()
cm = (A, MC.to_blob(A))
om = O.Model(cm[1])
for name, kw in (("gentle torques", dict(scale=0.1, seed=11, gravity_comp=True)), ("bench law", dict(scale=1.0, seed=5))):
    gq, oq, nprops, phys, gcen, ocen = _rollout_both(cm, om, N=N, T=200, flags=0, z_extra=0.0005, yaw=True,
                                                     solver="PGS", census=True, **kw)
    _divergence_report(f"PGS {name} ({N} envs)", gq, oq, nprops, gcen, ocen)
    err = np.abs(gq - oq)
    for i in range(N):
        err[:, i, 15 + 7 * int(nprops[i]):] = 0
    first = np.array([np.argmax(err[:, i].max(axis=1) > TOL) if (err[:, i].max() > TOL) else err.shape[0] for i in range(N)])
    print(f"   under 1e-4 at 250/500/750/1000 steps: {(first >= 250).mean():.3f} / {(first >= 500).mean():.3f} / {(first >= 750).mean():.3f} / {(first >= 1000).mean():.3f}; "
          f"per-env max: arm median {np.median(err[:, :, :7].max(axis=(0, 2))):.2e} max {err[:, :, :7].max():.2e}; fingers median {np.median(err[:, :, 7:15].max(axis=(0, 2))):.2e} max {err[:, :, 7:15].max():.2e}; "
          f"cubes median {np.median(err[:, :, 15:].max(axis=(0, 2))):.2e} max {err[:, :, 15:].max():.2e}", flush=True)
    phys.close()
