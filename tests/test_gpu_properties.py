"""Size-independent properties of the hot path at BASELINE.json's full batch (4096 envs per GPU,
configs[1] workload: 2-4 cubes per env, random actions per 5 ms tick).  The oracle cannot follow a
batch of this size in test time, so what is checked here are the invariants the domain offers:
bit-exact determinism, independence of an env from the batch it runs in, from how the rollout is cut
into launches and from how the env axis is sharded over ranks, and a finite, overflow-free state."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 4096
TICKS = 20


def _run(env_ids, cuts=(TICKS,), seed=0):
    import torch
    import bench
    from mujoco_robot_environments_amd import rng
    from mujoco_robot_environments_amd.physics import BatchedPhysics
    env_ids = np.asarray(env_ids)
    phys = BatchedPhysics(len(env_ids))
    bench.setup_envs(phys, seed, env_ids)
    acts = torch.from_numpy(rng.random_actions(seed, env_ids, np.arange(TICKS)).astype(np.float32)).to(phys.device)
    t = 0
    for c in cuts:
        phys.rollout(acts[t:t + c].contiguous(), control_steps=5)
        t += c
    assert t == TICKS
    out = (phys.qpos().copy(), phys.qvel().copy(), phys.status().copy(), phys.fallback_stats())
    phys.close()
    return out


@pytest.fixture(scope="module")
def full_batch():
    return _run(np.arange(N))


def test_full_batch_is_finite_and_within_capacity(full_batch):
    qpos, qvel, status, fb = full_batch
    assert np.isfinite(qpos).all() and np.isfinite(qvel).all()
    assert (status & 2).sum() == 0 and (status & 4).sum() == 0
    # cube quaternions stay normalised, cubes stay on the table (z of the centre above the table top)
    from mujoco_robot_environments_amd import rng
    nprops, _ = rng.prop_params(0, np.arange(N))
    for p in range(4):
        act = nprops > p
        qn = np.linalg.norm(qpos[act, 18 + 7 * p: 22 + 7 * p], axis=1)
        assert np.abs(qn - 1).max() < 1e-5
        assert (qpos[act, 17 + 7 * p] > 0.41).all()


def test_full_batch_is_deterministic(full_batch):
    again = _run(np.arange(N))
    assert np.array_equal(full_batch[0], again[0]) and np.array_equal(full_batch[1], again[1])
    assert np.array_equal(full_batch[2], again[2])


def test_env_does_not_depend_on_its_batch(full_batch):
    """The first 64 global env ids stepped alone give the same bits as inside the 4096 batch."""
    sub = _run(np.arange(64))
    assert np.array_equal(full_batch[0][:64], sub[0]) and np.array_equal(full_batch[1][:64], sub[1])


def test_rollout_does_not_depend_on_launch_cuts(full_batch):
    """20 ticks as one launch == 1 + 4 + 15 ticks as three launches (state round-trips through HBM)."""
    cut = _run(np.arange(N), cuts=(1, 4, 15))
    assert np.array_equal(full_batch[0], cut[0]) and np.array_equal(full_batch[1], cut[1])


def test_sharding_over_ranks_does_not_change_results(full_batch):
    """Rank r of 2 owns global ids [r*2048, (r+1)*2048): concatenated shards == the single batch."""
    a = _run(np.arange(0, N // 2))
    b = _run(np.arange(N // 2, N))
    assert np.array_equal(full_batch[0], np.concatenate([a[0], b[0]]))
    assert np.array_equal(full_batch[1], np.concatenate([a[1], b[1]]))
