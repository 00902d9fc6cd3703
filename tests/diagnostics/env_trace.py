"""Error growth of ONE env of a parity rollout (GPU diagnostic): per 25 steps the max |dq| of arm / fingers / cubes and
the constraint census (contacts + 64 * limit mask) of device and oracle.
    python tests/diagnostics/env_trace.py bench|gentle ENV [ENV ...]      (MRE_TRACE_SOLVER=PGS: the PGS path)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mujoco_robot_environments_amd.model import compile as MC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.test_gpu_parity import _rollout_both  # noqa: E402


def main():
    law = sys.argv[1]
    envs = [int(a) for a in sys.argv[2:]]
    A = MC.compile_scene()
    blob = MC.to_blob(A)
    om = O.Model(blob)
    kw = dict(scale=1.0, seed=5) if law == "bench" else dict(scale=0.1, seed=11, gravity_comp=True)
    gq, oq, nprops, phys, gcen, ocen = _rollout_both((A, blob), om, N=64, T=200, flags=0, z_extra=0.0005, yaw=True,
                                                     solver=os.environ.get("MRE_TRACE_SOLVER", "Newton"), census=True, **kw)
    err = np.abs(gq - oq)
    cnt, con = phys.contacts()
    names = A["_names"]["geoms"]
    for i in envs:
        print(f"env {i}: contacts at the end (geom pair, dist): " +
              ", ".join(f"{names[int(con[i, k, 0])]}-{names[int(con[i, k, 1])]} {con[i, k, 2]:.4f}" for k in range(abs(int(cnt[i]))) if con[i, k, 2] < 0.02 and not (names[int(con[i, k, 0])] == "table" and names[int(con[i, k, 1])].startswith("prop"))))
        n = 15 + 7 * int(nprops[i])
        print(f"env {i} ({int(nprops[i])} cubes): final max err arm {err[-1, i, :7].max():.2e} fingers {err[-1, i, 7:15].max():.2e}")
        for t in range(0, err.shape[0], 25):
            sl = slice(t, t + 25)
            print(f"  steps {t:4d}-{t + 24:4d}  arm {err[sl, i, :7].max():.2e}  fingers {err[sl, i, 7:15].max():.2e} "
                  f"(dof {7 + int(err[sl, i, 7:15].max(axis=0).argmax())})  cubes {err[sl, i, 15:n].max():.2e}  census dev "
                  f"{sorted(set((gcen[sl, i] & 0xFFFFFFFF).tolist()))} oracle {sorted(set((ocen[sl, i] & 0xFFFFFFFF).tolist()))}  q_fingers "
                  f"{np.round(oq[t, i, 7:15], 3).tolist()}")


if __name__ == "__main__":
    main()
