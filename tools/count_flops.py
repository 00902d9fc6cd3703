"""Algorithmic FLOPs per env-step of the bench workload, COUNTED on the CPU restatement ("to be
replaced by counted FLOPs from the instrumented CPU restatement").

The oracle's source is compiled a second time with `double` replaced by an operation-counting class
(oracle/flop_count.h, `make -C oracle flops`), so every + - * / the restatement executes is counted where it is
executed, per pipeline stage.  This script steps `nenv` envs of bench.py's workload (same scene, same seeded actions)
on that build and prints one JSON object: operations per env-step, per stage, for the tick windows asked for.

    python tools/count_flops.py [--nenv 16] [--windows 5:25,200:400] [--solver PGS|Newton|both]

What is counted is what MuJoCo's dense pipeline does on this model (nv = 39 < 60: MuJoCo keeps the constraint Jacobian
dense, so multiplications by structural zeros ARE part of the reference's work); the device's block-sparse kernels do
less.  Runs as a child process of bench.py's cpu_baseline leg (the counting library replaces the plain one through
MRE_ORACLE_LIB, which a process that already loaded the plain oracle cannot do)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nenv", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--windows", default="5:25,200:400")
    ap.add_argument("--solver", default="both")
    args = ap.parse_args()
    from oracle import oracle as O
    os.environ["MRE_ORACLE_LIB"] = O.build_flops()
    from mujoco_robot_environments_amd import placement, rng
    from mujoco_robot_environments_amd.model import compile as MC
    A = MC.compile_scene()
    m = O.Model(MC.to_blob(A))
    ids = np.arange(args.nenv)
    nprops, sizes = rng.prop_params(args.seed, ids)
    ws_min, ws_max = (0.35, -0.4, 0.43), (0.55, 0.4, 0.435)
    pose, _ = placement.sample_poses(args.seed, ids, nprops, sizes, ws_min, ws_max)
    windows = [tuple(int(x) for x in w.split(":")) for w in args.windows.split(",")]
    last = max(w[1] for w in windows)
    acts = rng.random_actions(args.seed, ids, np.arange(last))
    out = {"nenv": args.nenv, "stages": O.FLOP_STAGES, "unit": "operations per env-step (+ - * / = 1 each; sqrt / trig listed apart)",
           "mean_nv": float(np.mean(15 + 6 * nprops))}
    for solver in (["PGS", "Newton"] if args.solver == "both" else [args.solver]):
        envs = []
        for i in range(args.nenv):
            e = O.Env(m, int(nprops[i]), sizes[i])
            e.set_solver(solver)
            q = e.arr("qpos")
            q[:7] = A["home_qpos"]
            for p in range(int(nprops[i])):
                q[15 + 7 * p: 22 + 7 * p] = pose[i, p]
            e.freeze_robot(True)
            e.forward()
            e.step(300)                      # PropPlacer's settle, as bench.setup_envs
            e.freeze_robot(False)
            envs.append(e)
        O.flops_read(True)
        res = {}
        acc = np.zeros((len(O.FLOP_STAGES), 2))
        iters = 0
        for t in range(last):
            for i, e in enumerate(envs):
                e.arr("ctrl")[:] = acts[t, i]
                e.step(5)
                iters += e.solver_iters
            c = O.flops_read(True)
            for (a, b) in windows:
                if a <= t < b:
                    res.setdefault(f"{a}:{b}", np.zeros_like(acc))
                    res[f"{a}:{b}"] += c
        for k, v in res.items():
            a, b = (int(x) for x in k.split(":"))
            steps = args.nenv * (b - a) * 5
            res[k] = {"arith_per_env_step": float(v[:, 0].sum() / steps), "special_per_env_step": float(v[:, 1].sum() / steps),
                      "per_stage": {O.FLOP_STAGES[s]: round(float(v[s, 0] / steps), 1) for s in range(len(O.FLOP_STAGES))}}
        out[solver] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
