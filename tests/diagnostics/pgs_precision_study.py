"""CPU-only experiment behind VERDICT r3 item 1: what does a float32 PGS need to carry in fp64 so that its
1000-step trajectories stay within 1e-4 of the fp64 PGS oracle's?  (the PGS analogue of finger_precision_study.py)

    python tests/diagnostics/pgs_precision_study.py [N=64] [mode ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mujoco_robot_environments_amd.model import compile as MC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.diagnostics.finger_precision_study import run, TOL  # noqa: E402


def report(name, label, tr, ref, N):
    err = np.abs(tr - ref)[:, :, :43]
    switched = [(tr[:, i, 43] != ref[:, i, 43]).any() for i in range(N)]
    sw_at = [int(np.argmax(tr[:, i, 43] != ref[:, i, 43])) if switched[i] else -1 for i in range(N)]
    w = err.max(axis=2)
    first = np.array([np.argmax(w[:, i] > TOL) if w[:, i].max() > TOL else w.shape[0] for i in range(N)])
    unexpl = [i for i, f in enumerate(first) if f < w.shape[0] and (sw_at[i] < 0 or sw_at[i] > f)]
    if os.environ.get("ENVS"):
        for i in [int(x) for x in os.environ["ENVS"].split(",")]:
            print(f"   env {i}: fingers per 100 steps " + " ".join(f"{err[t:t + 100, i, 7:15].max():.1e}" for t in range(0, err.shape[0], 100)))
    fe = np.array([err[:, i, 7:15].max() for i in range(N) if not switched[i]])
    print(f"   finger error per env (no census switch): median {np.median(fe):.1e} p75 {np.quantile(fe, 0.75):.1e} p90 {np.quantile(fe, 0.9):.1e} "
          f"p97 {np.quantile(fe, 0.97):.1e} max {fe.max():.1e}")
    print(f"{name:6s} {label:28s}: under the bar {int((first >= w.shape[0]).sum())}/{N}, unexplained exits {len(unexpl)}; "
          f"max err arm {err[:, :, :7].max():.1e} fingers {err[:, :, 7:15].max():.1e} cubes {err[:, :, 15:].max():.1e}; "
          f"exits (env, step, switch at) {sorted([(i, int(f), sw_at[i]) for i, f in enumerate(first) if f < w.shape[0]], key=lambda x: x[1])[:10]}",
          flush=True)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
    want = [a for a in sys.argv[1:] if not a.isdigit()]
    O.build()
    om = O.Model(MC.to_blob(MC.compile_scene()))
    laws = (("gentle", (0.1, 11, True)), ("bench", (1.0, 5, False)))
    if os.environ.get("LAW"):
        laws = tuple(x for x in laws if x[0] == os.environ["LAW"])
    base = not os.environ.get("NOBASE")
    for name, law in laws:
        ref = run(om, law, N, "plain", solver="PGS")
        cases = [] if not base else [("all32 state", dict(mode="all")),
                 ("robot64 state", dict(mode="robot64")),
                 ("robot64 + arrays32 (r511)", dict(mode="robot64", rmask=511)),
                 ]
        for a in want:
            if a.startswith("p"):   # p<mask>: robot64 + r511 + PGS emulation mask (mro_set_pgs_emulation)
                m = int(a[1:])
                cases.append((f"robot64+r511+pgs{m}", dict(mode="robot64", rmask=511, hook=lambda e, m=m: e.pgs_emulation(m))))
        for label, kw in cases:
            tr = run(om, law, N, solver="PGS", **kw)
            report(name, label, tr, ref, N)


if __name__ == "__main__":
    main()
