"""SURVEY.md 8(f).4 on the GPU: ``BatchedControllerTuner.evaluate`` -- the population x replicate batch
that replaces the serial candidate loop of automated_controller_tuning/
rearrangement_controller_tuning.py:144-197 -- checked for the properties the reference's loop has:
the fitness is a pure function of the gains (same scenes for every candidate), signs are ignored
(``:188``), an unusable candidate earns the 1e6 penalty (``:177-183``), and one CMA-ES generation on
top of it runs end to end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OSC_YAML = [350.0, 20.0, 500.0, 100.0, 200.0, 30.0]     # kp, kd of position / orientation / nullspace
TUNED = [525.4, 215.2, 766.6, 158.6, 22.6, 7.2]          # config.TUNED_OSC_GAINS


def test_tuner_fitness_is_a_function_of_the_gains_only():
    from mujoco_robot_environments_amd.tuning import BatchedControllerTuner, CMAES
    pop, rep = 8, 8
    t = BatchedControllerTuner(popsize=pop, replicates=rep, seed=5)
    dead = [1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3]           # the arm never leaves home: every phase fails
    g = np.array([OSC_YAML, TUNED, OSC_YAML, [-x for x in TUNED], dead, TUNED, OSC_YAML, dead])
    f = t.evaluate(g)
    print("fitness:", np.round(f, 2).tolist())
    assert f.shape == (pop,) and np.isfinite(f).all()
    # every candidate saw the same scenes: equal gains (up to sign, :188) -> bit-equal fitness
    assert f[0] == f[2] == f[6]
    assert f[1] == f[3] == f[5]
    assert f[4] == f[7] == t.FAIL
    assert (~t.last["converged"][4]).all()
    # the penalty dominates any converged candidate's millimetre-scale reward
    assert f[1] < t.FAIL
    # a second evaluation draws new scenes (reset() advances the episode counter, like the reference's
    # env.reset() per candidate) -- again the same ones for every candidate
    f2 = t.evaluate(g)
    assert f2[0] == f2[2] == f2[6] and f2[1] == f2[3] == f2[5] and not np.array_equal(f, f2)
    # one generation of the reference's loop: ask -> evaluate(|x|) -> tell
    es = CMAES(np.array(TUNED), 20.0, pop, seed=0)
    x = es.ask()
    es.tell(x, t.evaluate(x))
    assert es.gen == 1 and np.isfinite(es.mean).all() and es.best_f <= t.FAIL
    t.close()
