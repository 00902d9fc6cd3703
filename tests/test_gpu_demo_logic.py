"""SURVEY.md 8(f).1 on the device: ``mre_sort_colours`` / ``mre_prop_place`` -- the reference's
``sort_colours`` / ``prop_pick`` / ``prop_place`` (tasks/rearrangement.py:579-751) as one launch pair
over the batch -- against the fp64 oracle replaying the reference's rule draw for draw: the cube is
moved to the sampled pose, physics.forward() evaluated, and the pose rejected while a detected contact
with a geom other than the table has dist <= 0.05 (:611-626)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(env, rng_np):
    """Cube 0 outside every zone with cube 1 sitting in cube 0's zone (so that place draws get
    rejected), the other cubes between the zones; every 16th env fully sorted, every other 16th with
    cube 0 already home; random yaws."""
    from mujoco_robot_environments_amd import demo_logic
    N = env.num_envs
    lo, hi = demo_logic.target_bounds(env._cfg.task, env.prop_colours, 4)
    q = env.physics.qpos().astype(np.float64)
    offs = [(-0.02, -0.06), (0.02, -0.02), (-0.02, 0.02), (0.02, 0.06)]
    free = [(0.45, 0.0), None, (0.45, -0.25), (0.45, 0.3)]   # x = 0.45 lies between the two zone rows
    for i in range(N):
        n = int(env.nprops[i])
        for p in range(n):
            a = 15 + 7 * p
            yaw = rng_np.uniform(-np.pi, np.pi)
            q[i, a + 3:a + 7] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
            z = 0.4 + env.prop_half_size[i, p, 2]
            own = 0.5 * (lo[i, p] + hi[i, p])
            c0 = 0.5 * (lo[i, 0] + hi[i, 0])
            if i % 16 == 15:
                xy = (own[0] + offs[p][0], own[1] + offs[p][1])
            elif p == 0:
                xy = (own[0] + offs[0][0], own[1] + offs[0][1]) if i % 16 == 7 else (0.45, 0.05 * (i % 3 - 1))
            elif p == 1:
                xy = (c0[0] + offs[3][0], c0[1] + offs[3][1])
            else:
                xy = free[p]
            q[i, a:a + 3] = [xy[0], xy[1], z]
    env.physics.set_state(qpos=q.astype(np.float32))
    return env.physics.qpos().astype(np.float64), lo, hi


def test_sort_colours_on_device_against_the_oracle(compiled_model, oracle_model):
    from mujoco_robot_environments_amd import demo_logic, rng
    from mujoco_robot_environments_amd.tasks.rearrangement import (BatchedRearrangementEnv, colour_separator_task_config,
                                                                  home_quat)
    from oracle import oracle as O
    A, _ = compiled_model
    N, seed = 64, 11
    env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=N, seed=seed, solver="Newton")
    env.reset()
    q0, lo, hi = _scene(env, np.random.default_rng(0))
    env._place_counts[:] = np.arange(N) % 3          # distinct RNG ticks per env
    counts = env._place_counts.copy()
    zones = np.concatenate([lo[..., :2], hi[..., :2]], axis=2)
    which, pick, place, att = env.physics.sort_colours(seed, counts, zones)
    assert np.array_equal(env.physics.qpos().astype(np.float64), q0), "the search must not touch the physics state"

    # selection and pick pose: the host restatement (numpy fp64)
    pos = q0[:, 15:43].reshape(N, 4, 7)[..., :3]
    quat = q0[:, 15:43].reshape(N, 4, 7)[..., 3:]
    hwhich = demo_logic.first_misplaced(pos, env.nprops, lo, hi)
    assert np.array_equal(which, hwhich)
    assert (which == -1).sum() >= 1 and (which == 0).sum() >= N // 2 and (which > 0).sum() >= 1
    hpick = demo_logic.batched_pick_pose(pos, quat, hwhich)
    sel = which >= 0
    assert np.abs(pick[sel] - hpick[sel]).max() < 1e-12
    assert (att[~sel] == 0).all() and (att[sel] > 0).all()

    # place: replay every draw on the oracle
    names = A["_names"]["geoms"]
    table = names.index("table")
    rejected = 0
    for i in np.nonzero(sel)[0]:
        p = int(which[i])
        gp = names.index(f"prop_{p}")
        e = O.Env(oracle_model, int(env.nprops[i]), env.prop_half_size[i].astype(np.float32).astype(np.float64))
        e.arr("qpos")[:43] = q0[i, :43]
        blo = np.array([lo[i, p, 0], lo[i, p, 1], 0.4])
        bhi = np.array([hi[i, p, 0], hi[i, p, 1], 0.4])
        for t in range(int(att[i])):
            u = rng.uniform(seed + 1, [env.env_ids[i]], [int(counts[i]) * 10000 + t], 3)[0, 0]
            cand = blo + (bhi - blo) * u
            e.arr("qpos")[15 + 7 * p:22 + 7 * p] = [*cand.astype(np.float32), 0.0, 0.0, 1.0, 0.0]
            e.forward()
            hit = any((int(c[13]) == gp or int(c[14]) == gp) and table not in (int(c[13]), int(c[14]))
                      and c[12] <= 0.05 for c in e.contacts())
            assert hit == (t < att[i] - 1), (i, t, att[i], [(int(c[13]), int(c[14]), c[12]) for c in e.contacts()])
            rejected += hit
        assert np.abs(place[i, :3] - cand).max() < 1e-15 and np.array_equal(place[i, 3:], [0.0, 0.0, 1.0, 0.0])
        assert (blo[:2] <= cand[:2]).all() and (cand[:2] <= bhi[:2]).all()
    print(f"draws per env: mean {att[sel].mean():.2f} max {att[sel].max()}, {rejected} rejections replayed on the oracle")
    assert rejected > 20, "the rejection branch must have been exercised"

    # the env-level call: same poses, home pose for idle envs, RNG call counters advance
    env._place_counts[:] = counts
    prog, epick, eplace = env.sort_colours()
    assert np.array_equal(prog, sel)
    assert np.array_equal(epick[sel], pick[sel]) and np.array_equal(eplace[sel], place[sel])
    assert np.allclose(epick[~sel, 3:], home_quat()) and np.allclose(eplace[~sel, :3], np.atleast_2d(env.eef_home_pose)[0])
    assert np.array_equal(env._place_counts, counts + sel)
    env.close()


def test_prop_place_single_env_reference_signature(compiled_model):
    """prop_place(prop_id, min_pose, max_pose) of the batch-of-one env: inside the bounds, clear of the other
    cubes by the reference's 0.05, a new draw on every call, and an impossible request raises."""
    from mujoco_robot_environments_amd.tasks.rearrangement import RearrangementEnv, colour_separator_task_config, PROP_GEOM_ID0
    env = RearrangementEnv(cfg=colour_separator_task_config(), render=False)
    env.reset()
    info = env.props_info
    ids = sorted(info.keys())
    others = np.array([info[k]["position"] for k in ids[1:]])
    ws = env._cfg.task.initializers.workspace
    a = env.prop_place(ids[0])
    b = env.prop_place(ids[0])
    assert a.shape == (7,) and not np.array_equal(a[:3], b[:3])
    for pose in (a, b):
        assert (np.asarray(ws.min_pose) <= pose[:3]).all() and (pose[:3] <= np.asarray(ws.max_pose)).all()
        # box surfaces further apart than 0.05 -> centres further apart than 0.05 + the cube's edge (3-D; the
        # sampled pose hovers ~1.5 cm above the resting cubes)
        assert (np.linalg.norm(others[:, :2] - pose[:2], axis=1) > 0.075).all()
    # a box the size of a cube around another cube's centre: every pose touches it
    c = others[0]
    with pytest.raises(Exception, match="Failed to find collision free place pose"):
        env.prop_place(ids[0], [c[0] - 0.01, c[1] - 0.01, 0.4], [c[0] + 0.01, c[1] + 0.01, 0.4])
    assert ids[0] == PROP_GEOM_ID0
    env.close()
