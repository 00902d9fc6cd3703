"""Pyramidal cones on the rearrangement scene (NOT what the reference runs there -- 2f85.xml sets elliptic -- but the
scene with the most contact structure: cube-cube, finger-cube, arm-cube, ground): device vs oracle, both solvers.
usage: python tests/diagnostics/pyramidal_sample.py [nenvs=32] [ticks=100]   (needs the GPU; not a test)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mujoco_robot_environments_amd.model import compile as MC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.test_gpu_newton import _divergence_report  # noqa: E402
from tests.test_gpu_parity import _rollout_both  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
A = MC.compile_scene()
A["opt_cone"][:] = 0
cm = (A, MC.to_blob(A))
om = O.Model(cm[1])
for solver in ("Newton", "PGS"):
    for name, kw in (("gentle", dict(scale=0.1, seed=11, gravity_comp=True)), ("bench law", dict(scale=1.0, seed=5))):
        gq, oq, nprops, phys, gcen, ocen = _rollout_both(cm, om, N=N, T=T, flags=0, z_extra=0.0005, yaw=True,
                                                         solver=solver, census=True, **kw)
        _divergence_report(f"pyramidal {solver} {name}", gq, oq, nprops, gcen, ocen)
        err = np.abs(gq - oq)
        for i in range(N):
            err[:, i, 15 + 7 * int(nprops[i]):] = 0
        print(f"   status {np.bincount(phys.status().astype(np.int64))} arm med {np.median(err[:, :, :7].max(axis=(0, 2))):.2e} "
              f"cubes med {np.median(err[:, :, 15:].max(axis=(0, 2))):.2e} iters {phys.solver_stats()[:, 2].mean():.2f}", flush=True)
        phys.close()
