#!/usr/bin/env python3
"""SURVEY.md 8(f).4 -- the reference's CMA-ES controller tuning
(mujoco_robot_environments/automated_controller_tuning/rearrangement_controller_tuning.py:144-197)
with the whole population evaluated as ONE batch: candidate c runs on envs [c*R, (c+1)*R) (R scene
replicates shared by all candidates), per-env OSC gains are kernel inputs, fitness = the reference's
1000 x placement distance (1e6 when a scripted phase fails to converge).

    python examples/controller_tuning.py --popsize 256 --replicates 16 --generations 5
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mujoco_robot_environments_amd.tuning import CMAES, BatchedControllerTuner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--popsize", type=int, default=256)
    ap.add_argument("--replicates", type=int, default=16)
    ap.add_argument("--generations", type=int, default=5)
    ap.add_argument("--sigma", type=float, default=100.0, help="reference: sigma_init=500 around 500")
    args = ap.parse_args()
    tuner = BatchedControllerTuner(args.popsize, args.replicates)
    x0 = np.array([350.0, 20.0, 500.0, 100.0, 200.0, 30.0])  # osc.yaml:5-15
    es = CMAES(x0, args.sigma, args.popsize)
    base = tuner.evaluate(np.tile(x0, (args.popsize, 1)))
    print(f"reference gains {x0}: fitness {base[0]:.1f} "
          f"(converged in {tuner.last['converged'][0].mean():.0%} of replicates, "
          f"median placement error {np.median(tuner.last['dist'][0]) * 100:.1f} cm)")
    t0 = time.time()
    for g in range(args.generations):
        x = es.ask()
        x[:, 1::2] = np.clip(np.abs(x[:, 1::2]), 1.0, None)  # keep some damping
        fit = tuner.evaluate(x)
        es.tell(x, fit)
        b = int(np.argmin(fit))
        print(f"generation {g}: best fitness {fit[b]:.1f} (all-time {es.best_f:.1f}) gains {np.abs(x[b]).round(1)} | "
              f"converged {tuner.last['converged'][b].mean():.0%}, median placement error "
              f"{np.median(tuner.last['dist'][b]) * 100:.1f} cm | "
              f"{(g + 1) * tuner.num_envs * 18000 / (time.time() - t0):.3g} env-steps/s")
    print("best gains", np.abs(es.best_x).round(1), "fitness", es.best_f)
    tuner.close()


if __name__ == "__main__":
    main()
