"""Massively-parallel controller tuning: the CMA-ES loop of the reference's
``automated_controller_tuning/rearrangement_controller_tuning.py:144-197`` with the whole population
(times a few scene replicates) evaluated as ONE batch of environments -- per-env OSC gains are kernel
inputs (``mre_osc_configure_env``).  evosax / jax are not needed: ``CMAES`` below is the standard
(mu/mu_w, lambda)-CMA-ES (Hansen, "The CMA Evolution Strategy: A Tutorial") in numpy."""
from __future__ import annotations

import numpy as np


class CMAES:
    """ask() -> [popsize, n] candidates; tell(x, fitness) (minimisation)."""

    def __init__(self, x0, sigma0: float, popsize: int, seed: int = 0):
        self.n = n = len(x0)
        self.mean = np.asarray(x0, np.float64).copy()
        self.sigma = float(sigma0)
        self.lam = int(popsize)
        self.mu = self.lam // 2
        w = np.log(self.mu + 0.5) - np.log(np.arange(1, self.mu + 1))
        self.w = w / w.sum()
        self.mueff = 1.0 / np.sum(self.w ** 2)
        self.cc = (4 + self.mueff / n) / (n + 4 + 2 * self.mueff / n)
        self.cs = (self.mueff + 2) / (n + self.mueff + 5)
        self.c1 = 2 / ((n + 1.3) ** 2 + self.mueff)
        self.cmu = min(1 - self.c1, 2 * (self.mueff - 2 + 1 / self.mueff) / ((n + 2) ** 2 + self.mueff))
        self.damps = 1 + 2 * max(0.0, np.sqrt((self.mueff - 1) / (n + 1)) - 1) + self.cs
        self.pc = np.zeros(n)
        self.ps = np.zeros(n)
        self.C = np.eye(n)
        self.chi_n = np.sqrt(n) * (1 - 1 / (4 * n) + 1 / (21 * n * n))
        self.gen = 0
        self.rng = np.random.default_rng(seed)
        self.best_x, self.best_f = self.mean.copy(), np.inf
        self._eig()

    def _eig(self):
        self.C = np.triu(self.C) + np.triu(self.C, 1).T
        d2, self.B = np.linalg.eigh(self.C)
        self.D = np.sqrt(np.maximum(d2, 1e-20))

    def ask(self) -> np.ndarray:
        z = self.rng.standard_normal((self.lam, self.n))
        self._y = (z * self.D) @ self.B.T
        return self.mean + self.sigma * self._y

    def tell(self, x: np.ndarray, fitness) -> None:
        f = np.asarray(fitness, np.float64)
        order = np.argsort(f, kind="stable")
        if f[order[0]] < self.best_f:
            self.best_f, self.best_x = float(f[order[0]]), np.array(x[order[0]])
        y = (np.asarray(x)[order[:self.mu]] - self.mean) / self.sigma
        yw = self.w @ y
        self.mean = self.mean + self.sigma * yw
        inv_sqrt = self.B @ np.diag(1.0 / self.D) @ self.B.T
        self.ps = (1 - self.cs) * self.ps + np.sqrt(self.cs * (2 - self.cs) * self.mueff) * (inv_sqrt @ yw)
        self.gen += 1
        hs = np.linalg.norm(self.ps) / np.sqrt(1 - (1 - self.cs) ** (2 * self.gen)) / self.chi_n < 1.4 + 2 / (self.n + 1)
        self.pc = (1 - self.cc) * self.pc + hs * np.sqrt(self.cc * (2 - self.cc) * self.mueff) * yw
        self.C = ((1 - self.c1 - self.cmu) * self.C
                  + self.c1 * (np.outer(self.pc, self.pc) + (1 - hs) * self.cc * (2 - self.cc) * self.C)
                  + self.cmu * (y.T * self.w) @ y)
        self.sigma *= np.exp((self.cs / self.damps) * (np.linalg.norm(self.ps) / self.chi_n - 1))
        self._eig()


class BatchedControllerTuner:
    """Fitness of a population of OSC gain sets on the scripted pick + place, one batch per
    generation.  Env e evaluates candidate e // replicates on scene replicate e % replicates
    (different cube layouts, the same layouts for every candidate), the candidate's fitness is the
    mean over its replicates of the reference's reward (rearrangement_controller_tuning.py:118-123:
    1000 x distance of the moved cube to the place target) with the reference's 1e6 penalty when a
    scripted phase does not converge (``:177-183``)."""

    FAIL = 1e6

    def __init__(self, popsize: int = 256, replicates: int = 16, cfg=None, seed: int = 0):
        from .tasks.rearrangement import BatchedRearrangementEnv, colour_separator_task_config
        self.popsize, self.replicates = int(popsize), int(replicates)
        self.num_envs = self.popsize * self.replicates
        # every candidate sees the same `replicates` scenes: env ids repeat with period `replicates`
        ids = np.tile(np.arange(self.replicates), self.popsize)
        self.env = BatchedRearrangementEnv(cfg=cfg or colour_separator_task_config(), num_envs=self.num_envs,
                                           seed=seed, env_ids=ids)

    def evaluate(self, gains: np.ndarray) -> np.ndarray:
        """gains [popsize, 6] -> fitness [popsize]."""
        g = np.abs(np.asarray(gains, np.float64))  # "only consider positive params" (:188)
        assert g.shape == (self.popsize, 6)
        env = self.env
        env.reset()
        env._robot.arm_controller.set_gains_per_env(np.repeat(g, self.replicates, axis=0))
        in_progress, pick, place = env.sort_colours()
        pos0 = env.physics.sites()[2]
        which = np.array([int(np.argmin(np.linalg.norm(pos0[i, :env.nprops[i], :2] - pick[i, :2], axis=1)))
                          for i in range(self.num_envs)])
        env.failed_phase[:] = ""
        env.step({"pose": pick.copy()})
        env.step({"pose": place.copy()})
        pos1 = env.physics.sites()[2]
        dist = np.linalg.norm(pos1[np.arange(self.num_envs), which, :2] - place[:, :2], axis=1)
        fit = 1000.0 * dist
        fit[~env.last_converged] = self.FAIL
        fit[~in_progress] = 0.0
        self.last = dict(dist=dist.reshape(self.popsize, self.replicates),
                         converged=env.last_converged.reshape(self.popsize, self.replicates))
        return fit.reshape(self.popsize, self.replicates).mean(axis=1)

    def close(self):
        self.env.close()
