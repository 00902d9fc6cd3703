"""GPU tests of the drop-in Python API: reset()/step() 4-tuples and observation shapes of the
reference (tasks/rearrangement.py:297-356,442-478), batch-of-1 == reference shapes, scripted
pick through the fused run_controller launches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_env_reference_shapes():
    from mujoco_robot_environments_amd.tasks.rearrangement import RearrangementEnv, StepType, colour_separator_task_config
    env = RearrangementEnv(viewer=False, cfg=colour_separator_task_config())
    step_type, reward, discount, obs = env.reset()
    assert step_type == StepType.FIRST and reward == 0.0 and discount == 0.0
    assert obs["overhead_camera/rgb"].shape == (480, 640, 3) and obs["overhead_camera/rgb"].dtype == np.uint8
    assert obs["overhead_camera/depth"].shape == (480, 640) and obs["overhead_camera/depth"].dtype == np.float32
    assert set(env.action_spec()) == {"pose", "pixel_coords", "gripper_rot"}
    assert env.observation_spec()["overhead_camera/rgb"].shape == (480, 640, 3)
    assert env.eef_home_pose.shape == (3,) and abs(env.eef_home_pose[0] - (0.30702 - 0.1)) < 2e-3
    info = env.props_info
    assert 2 <= len(info) <= 4
    for a in info.values():
        assert abs(a["position"][2] - 0.4155) < 2e-3, "cubes settle on the table top"
        assert a["bbox"].shape == (4,)
    in_progress, pick, place = env.sort_colours()
    assert in_progress and pick.shape == (7,) and place.shape == (7,)
    px = env.world_2_pixel("overhead_camera/overhead_camera", pick[:3])
    assert px.shape == (2,) and px.dtype == np.int32 and 0 <= px[0] < 640 and 0 <= px[1] < 480
    back = env.pixel_2_world("overhead_camera/overhead_camera", px.astype(np.float64))
    assert np.linalg.norm(back[:2] - pick[:2]) < 0.01
    md = env.get_camera_metadata()
    assert set(md) == {"intrinsics", "extrinsics"} and md["intrinsics"]["cx"] == (640 - 1) / 2
    env.close()


def test_batched_scripted_pick_runs_all_phases():
    from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, StepType, colour_separator_task_config
    N = 4
    env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=N)
    ts = env.reset()
    assert ts.observation["overhead_camera/rgb"].shape == (N, 480, 640, 3)
    prog, pick, place = env.sort_colours()
    assert prog.shape == (N,) and pick.shape == (N, 7) and place.shape == (N, 7)
    p = pick.copy()
    ts = env.step({"pose": p, "pixel_coords": env.world_2_pixel("overhead_camera/overhead_camera", pick[:, :3]),
                   "gripper_rot": 0.0})
    assert ts.step_type == StepType.MID and env.mode == "place"
    assert np.allclose(p[:, 2], 0.575), "pick() overwrites pose[2] in place (reference :362)"
    nsteps = round(env._robot.time / 0.001)
    assert 300 + 9000 <= nsteps <= 2000 + 9000 + 25, "settle (0.3..2 s) + five phases of 2+2+1+2+2 s"
    assert env.last_converged.all(), list(env.failed_phase)
    tcp, eef, props = env.physics.sites()
    # reference quirk: eef_home_pose is read from the PINCH site but used as target of the
    # CONTROLLER (attachment) site (tasks/rearrangement.py:326-327,394-397)
    assert np.abs(eef[:, :3] - env.eef_home_pose).max() < 0.01, "controller site is back at the home target"
    assert (env.physics.status() & 2 == 0).all()
    lifted = [(props[i, :int(env.nprops[i]), 2] > 0.5).any() for i in range(N)]
    print("cubes held after pick:", sum(lifted), "/", N)
    env.close()


def test_place_props_follows_prop_placer_against_the_oracle(compiled_model, oracle_model):
    """mre_place_props = PropPlacer.__call__ (environment/prop_initializer.py:164-283), checked against the
    ORACLE running the reference's algorithm: props one after the other, pose ~ U(workspace) with the
    counter RNG, physics.forward(), rejected while the prop has a detected contact (dist < margin) with
    any geom but the table.  Poses must agree draw for draw (same attempts accepted).  Then the settle:
    per-env exit on max|qvel| < 1e-3, max|qacc| < 1e-2, t > 0.3 s with the robot frozen; step counts
    against the oracle's, and a mask restricts everything to the envs being reset."""
    from mujoco_robot_environments_amd import rng
    from mujoco_robot_environments_amd.physics import BatchedPhysics
    from oracle import oracle as O
    A, _ = compiled_model
    N, off, seed, max_attempts = 64, 4096, 5, 1000
    ids = np.arange(off, off + N)
    nprops, sizes = rng.prop_params(seed, ids)
    sizes = sizes.astype(np.float32).astype(np.float64)
    lo = np.array([0.35, -0.4, 0.43], np.float32)
    hi = np.array([0.55, 0.4, 0.435], np.float32)
    phys = BatchedPhysics(N, model=A, solver="Newton")
    phys.set_env_id_offset(off)
    phys.set_props(nprops, sizes)
    phys.reset()
    phys.place_props(seed, lo, hi, settle_steps=0)
    q = phys.qpos()
    geom_names = A["_names"]["geoms"]
    table = geom_names.index("table")
    nattempts = []
    envs = []
    for i in range(N):
        e = O.Env(oracle_model, int(nprops[i]), sizes[i])
        e.set_solver("Newton")
        e.reset()   # cubes parked out of reach, like the library's start state
        e.arr("qpos")[:7] = A["home_qpos"]
        for p in range(int(nprops[i])):
            gp = geom_names.index(f"prop_{p}")
            for att in range(max_attempts):
                u = rng.uniform(seed, [ids[i]], [p * max_attempts + att], 4)[0, 0]
                pos = (lo.astype(np.float64) + (hi.astype(np.float64) - lo.astype(np.float64)) * u[:3]).astype(np.float32)
                yaw = np.pi * u[3]
                e.arr("qpos")[15 + 7 * p: 22 + 7 * p] = [*pos, np.float32(np.cos(yaw / 2)), 0, 0, np.float32(np.sin(yaw / 2))]
                e.forward()
                hit = any((int(c[13]) == gp or int(c[14]) == gp) and table not in (int(c[13]), int(c[14])) for c in e.contacts())
                if not hit:
                    break
            else:
                raise AssertionError("oracle could not place")
            nattempts.append(att + 1)
        n = int(nprops[i])
        assert np.abs(q[i, 15:15 + 7 * n] - e.arr("qpos")[15:15 + 7 * n]).max() < 1e-6, (i, q[i, 15:15 + 7 * n], e.arr("qpos")[15:15 + 7 * n])
        envs.append(e)
    print("attempts per prop: mean %.2f max %d" % (np.mean(nattempts), max(nattempts)))
    assert max(nattempts) > 1, "the rejection branch must have been exercised"
    # contact accessor: what the rejection test read
    cnt, con = phys.contacts()
    for i in (0, 7, 31):
        oc = envs[i]
        oc.forward()
        ol = sorted((int(c[13]), int(c[14]), round(float(c[12]), 5)) for c in oc.contacts())
        gl = sorted((int(con[i, k, 0]), int(con[i, k, 1]), round(float(con[i, k, 2]), 5)) for k in range(abs(int(cnt[i]))))
        assert [x[:2] for x in ol] == [x[:2] for x in gl], (ol, gl)
        assert max(abs(a[2] - b[2]) for a, b in zip(ol, gl)) < 2e-5
    # settle: the even envs only, per-env exit
    mask = (np.arange(N) % 2 == 0).astype(np.uint8)
    before = phys.qpos().copy()
    phys.place_props(seed + 1, lo, hi, mask=mask, settle_steps=300)
    after = phys.qpos()
    gsteps = phys.settle_steps()
    assert np.array_equal(after[mask == 0], before[mask == 0]), "unmasked envs must not move"
    for i in np.nonzero(mask)[0]:
        n = int(nprops[i])
        z = after[i, 15:15 + 7 * n].reshape(n, 7)[:, 2]
        assert np.abs(z - (0.4 + sizes[i, :n, 2])).max() < 1e-3, "cubes rest on the table top"
    assert (np.abs(gsteps[mask == 1]) >= 300).all() and (gsteps[mask == 1] > 0).all(), gsteps[mask == 1][:8]
    assert (gsteps[mask == 1] < 2000).all()
    print("settle steps (device): min %d median %d max %d" % (gsteps[mask == 1].min(), np.median(gsteps[mask == 1]), gsteps[mask == 1].max()))
    assert np.abs(phys.qvel()[mask == 1][:, 15:]).max() < 1e-3
    assert (phys.status() == 0).all()


def test_settle_exit_matches_oracle(compiled_model, oracle_model):
    """The per-env settle exit (prop_initializer.py:240-258) against the oracle applying the same rule
    from the same placed poses: step counts within two steps, final cube poses at the bar."""
    from mujoco_robot_environments_amd import rng
    from mujoco_robot_environments_amd.physics import BatchedPhysics
    from oracle import oracle as O
    A, _ = compiled_model
    N, seed = 32, 8
    ids = np.arange(N)
    nprops, sizes = rng.prop_params(seed, ids)
    sizes = sizes.astype(np.float32).astype(np.float64)
    lo = np.array([0.35, -0.4, 0.43], np.float32)
    hi = np.array([0.55, 0.4, 0.435], np.float32)
    phys = BatchedPhysics(N, model=A, solver="Newton")
    phys.set_props(nprops, sizes)
    phys.reset()
    phys.place_props(seed, lo, hi, settle_steps=0)
    q0 = phys.qpos().copy()
    phys.set_state(q0, np.zeros((N, 39), np.float32))
    # settle through the library: same poses (mask = all, but poses re-drawn with the same seed)
    phys.reset()
    phys.place_props(seed, lo, hi, settle_steps=300)
    gsteps = phys.settle_steps()
    gq = phys.qpos()
    osteps = np.zeros(N, int)
    worst = 0.0
    for i in range(N):
        n = int(nprops[i])
        e = O.Env(oracle_model, n, sizes[i])
        e.set_solver("Newton")
        e.arr("qpos")[:43] = q0[i]
        e.freeze_robot(True)
        e.forward()
        for k in range(2000):
            e.step(1)
            v = np.abs(e.arr("qvel")[15:15 + 6 * n]).max()
            a = np.abs(e.arr("qacc")[15:15 + 6 * n]).max()
            if v < 1e-3 and a < 1e-2 and k + 1 > 300:
                break
        osteps[i] = k + 1
        if abs(osteps[i] - gsteps[i]) <= 2:
            worst = max(worst, float(np.abs(gq[i, 15:15 + 7 * n] - e.arr("qpos")[15:15 + 7 * n]).max()))
    close = np.abs(osteps - gsteps) <= 2
    print("settle steps device", gsteps[:10].tolist(), "oracle", osteps[:10].tolist(), "within two steps:", close.mean(),
          "pose err of those %.1e" % worst)
    assert close.mean() >= 0.9
    assert worst < 1e-4


def test_rendered_observations_bboxes_and_pixel_round_trip():
    """With rendering on, observations are the overhead camera's images, props_info boxes come from
    the segmentation image (tasks/rearrangement.py:254-280) and agree with the projected cube corners
    to a pixel or two, and pixel_2_world(world_2_pixel(p)) returns to the cube's top face
    (tasks/rearrangement.py:500-548)."""
    from mujoco_robot_environments_amd.tasks.rearrangement import (BatchedRearrangementEnv, RearrangementEnv,
                                                                    colour_separator_task_config, OVERHEAD)
    env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=4, render=True)
    ts = env.reset()
    poses = env.physics.sites()[2]
    boxes = env.prop_bboxes()
    ts = env.step({"pose": env.sort_colours()[1]})   # the observation is rendered BEFORE acting
    rgb, depth = ts.observation["overhead_camera/rgb"], ts.observation["overhead_camera/depth"]
    assert tuple(rgb.shape) == (4, 480, 640, 3) and str(rgb.dtype) == "torch.uint8"
    assert tuple(depth.shape) == (4, 480, 640) and float(depth.min()) > 0.3 and float(depth.max()) <= 1.3 + 1e-3
    for i in range(4):
        info_seg = env.props_info_env(i, poses, boxes)
        env.render_observations = False
        info_geo = env.props_info_env(i, poses)
        env.render_observations = True
        for k in info_seg:
            assert len(info_seg[k]["bbox"]) == 4
            assert np.abs(info_seg[k]["bbox"] - info_geo[k]["bbox"]).max() <= 2, (info_seg[k]["bbox"], info_geo[k]["bbox"])
            assert 0.0 <= info_seg[k]["rgba"].min() and info_seg[k]["rgba"].max() <= 1.0
    env.reset()
    poses = env.physics.sites()[2]
    for i in range(4):
        p = int(env.nprops[i]) - 1
        top = poses[i, p, :3].astype(np.float64) + [0, 0, env.prop_half_size[i, p, 2]]
        px = env.world_2_pixel(OVERHEAD, top)
        back = env.pixel_2_world(OVERHEAD, np.asarray(px, np.float64).reshape(-1)[:2], env=i)
        assert np.abs(back - top).max() < 3e-3, (back, top)
    env.close()
    one = RearrangementEnv(cfg=colour_separator_task_config())
    ts = one.reset()
    obs = one._compute_observation()
    assert obs["overhead_camera/rgb"].shape == (480, 640, 3) and obs["overhead_camera/rgb"].dtype == np.uint8
    assert obs["overhead_camera/depth"].shape == (480, 640) and obs["overhead_camera/depth"].dtype == np.float32
    assert len(one.props_info) == int(one.nprops[0])
    one.close()


def test_interactive_tuning_tick_follows_the_mocap_target_like_the_oracle(compiled_model, oracle_model):
    """tasks/rearrangement.py:753-779: target = mocap pose + (0, 0, 0.175), one command held for 5 steps per
    call.  8 envs with different mocap poses, 300 calls, against the oracle running the same tick loop."""
    from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, colour_separator_task_config, home_quat
    from oracle import oracle as O
    N, T = 8, 300
    env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=N, seed=2, solver="Newton")
    env.reset()
    q0 = env.physics.qpos().astype(np.float64)
    assert np.allclose(env.mocap_pos, [0.4, 0.0, 0.6]) and np.allclose(env.mocap_quat, home_quat())
    mp = np.array([[0.4 + 0.02 * (i % 3), -0.1 + 0.03 * i, 0.55 + 0.01 * i] for i in range(N)])
    t0 = env._robot.time
    for _ in range(T):
        env.interactive_tuning(mocap_pos=mp)
    assert abs(env._robot.time - t0 - T * 5 * 0.001) < 1e-9 and not env.time_limit_exceeded()
    gq = env.physics.qpos().astype(np.float64)
    _, eef, _ = env.physics.sites()
    assert np.abs(eef[:, :3] - (mp + [0.0, 0.0, 0.175])).max() < 0.02, "the controller site tracks the mocap target"
    worst = 0.0
    for i in range(N):
        e = O.Env(oracle_model, int(env.nprops[i]), env.prop_half_size[i].astype(np.float32).astype(np.float64))
        e.set_solver("Newton")
        e.arr("qpos")[:43] = q0[i, :43]
        e.forward()
        p = O.make_osc()
        p.target_pos[:] = mp[i] + [0.0, 0.0, 0.175]
        p.target_quat[:] = home_quat()
        for _ in range(T):
            e.run_controller(p, 0.0, 1, 5)
        n = int(env.nprops[i])
        worst = max(worst, float(np.abs(gq[i, :15 + 7 * n] - e.arr("qpos")[:15 + 7 * n]).max()))
    print(f"interactive_tuning: {N} envs x {T} ticks, max |dq| vs oracle {worst:.2e}")
    assert worst < 1e-4
    env.close()
