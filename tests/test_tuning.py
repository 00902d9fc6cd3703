"""CPU tests of the tuning support: the numpy CMA-ES that replaces evosax in the
batched form of automated_controller_tuning/rearrangement_controller_tuning.py:144-197."""
import numpy as np

from mujoco_robot_environments_amd.tuning import CMAES


def _run(f, x0, sigma, popsize, gens, seed=0):
    es = CMAES(x0, sigma, popsize, seed=seed)
    for _ in range(gens):
        x = es.ask()
        es.tell(x, [f(v) for v in x])
    return es


def test_cmaes_minimises_sphere():
    es = _run(lambda v: float(np.sum((v - 3.0) ** 2)), np.zeros(6), 1.0, 20, 120)
    assert es.best_f < 1e-8 and np.abs(es.best_x - 3.0).max() < 1e-3


def test_cmaes_adapts_to_an_ill_conditioned_rotated_quadratic():
    rng = np.random.default_rng(1)
    q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
    d = np.logspace(0, 4, 6)
    es = _run(lambda v: float(((q @ v) ** 2 * d).sum()), np.full(6, 2.0), 1.0, 24, 400)
    assert es.best_f < 1e-8
    assert es.sigma < 1e-2  # step size has contracted


def test_cmaes_same_shapes_and_positive_gain_convention():
    es = CMAES(np.full(6, 500.0), 500.0, 20)  # the reference's init (init_min = init_max = 500, sigma 500)
    x = es.ask()
    assert x.shape == (20, 6)
    es.tell(x, np.abs(x).sum(axis=1))  # fitness is evaluated on |x| in the reference (:188)
    assert np.isfinite(es.mean).all() and es.gen == 1
