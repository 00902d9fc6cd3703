"""CPU tests of the host-side logic: model compiler, config composer, RNG keyed by
global env id, placement, tick counting of run_controller."""
import numpy as np
import pytest

from mujoco_robot_environments_amd import config, placement, rng
from mujoco_robot_environments_amd.model import compile as MC


def test_scene_dimensions(compiled_model):
    A, blob = compiled_model
    assert int(A["nbody"][0]) == 20 and int(A["nv"][0]) == 39 and int(A["nq"][0]) == 43
    assert int(A["nu"][0]) == 8 and int(A["npair"][0]) == 84 and int(A["neq"][0]) == 3
    # robot dofs precede the cubes, arm dofs are 0..6 (kernel assumption)
    assert list(A["arm_dof"]) == list(range(7))
    assert list(A["body_dofadr"][16:20]) == [15, 21, 27, 33]
    assert blob[:4] == (0x4D524542).to_bytes(4, "little")


def test_fused_inertias_are_physical(compiled_model):
    A, _ = compiled_model
    I = A["body_inertia"][1:]
    assert (I > 0).all()
    s = np.sort(I, axis=1)
    assert (s[:, 2] <= (s[:, 0] + s[:, 1]) * (1 + 1e-6)).all(), "triangle inequality (MuJoCo compiler check)"
    # link7 carries the fused attachment + gripper base (0.735522 + 0.777441 kg)
    assert abs(A["body_mass"][7] - (0.735522 + 0.777441)) < 1e-7   # (constants are rounded to float32)


def test_home_kinematics_known_answer(compiled_model):
    """Panda flange at the 'home' keyframe: x = 0.307, z = 0.59 above the base (0.4)."""
    A, _ = compiled_model
    q = A["qpos0"].copy()
    q[:7] = A["home_qpos"]
    xpos, xquat = MC.forward_kinematics(A, q)
    s = int(A["eef_site"][0])
    b = int(A["site_bodyid"][s])
    p = xpos[b] + MC.qrot(xquat[b], A["site_pos"][s])
    assert np.allclose(p, [0.30702, 0.0, 0.4 + 0.59027], atol=2e-4)


def test_connect_anchors_close_the_linkage(compiled_model):
    A, _ = compiled_model
    xpos, xquat = MC.forward_kinematics(A, A["qpos0"])
    for e in range(2):
        b1, b2 = A["eq_obj"][e]
        p1 = xpos[b1] + MC.qrot(xquat[b1], A["eq_data"][e, :3])
        p2 = xpos[b2] + MC.qrot(xquat[b2], A["eq_data"][e, 3:6])
        assert np.allclose(p1, p2, atol=1e-12)


def test_config_compose_matches_reference_tree():
    c = config.default_config()
    assert c.physics_dt == 0.001 and c.control_dt == 0.005 and c.gravity == [0.0, 0.0, -9.8]
    assert c.arena.props.min_objects == 2 and c.arena.props.max_objects == 5
    cp = c.robots.arm.controller_config.controller_params
    assert cp.physics_dt == 0.001 and cp.control_dt == 0.005  # ${...} interpolation
    assert cp.gains.position.kp == 350.0 and cp.convergence.orientation_threshold == 68e-3
    assert c.robots.arm.controller_config.controller.controller_config.gains.nullspace.kd == 30.0
    assert c.task.initializers.workspace.min_pose == [0.35, -0.4, 0.43]
    t = config.colour_separator_task_config()
    assert t.name == "colour_splitter" and t.task.colour_target_map.green == "bottom_left"
    assert t.task.target_locations.top_left.size == [0.075, 0.15, 0.01]
    with pytest.raises(config.MissingConfig):
        config.compose("rearrangement", [])  # props has no default in the reference tree
    with pytest.raises(KeyError):
        config.compose("rearrangement", ["arena/props=colour_splitter", "nonexistent=1"])
    c2 = config.compose("rearrangement", ["arena/props=colour_splitter", "physics_dt=0.002"])
    assert c2.robots.arm.controller_config.controller_params.physics_dt == 0.002
    assert c.physics_dt == 0.001, "compose must be re-entrant (no shared state)"


def test_rng_is_a_pure_function_of_global_env_id():
    full = rng.random_actions(3, np.arange(16), np.arange(5))
    lo = rng.random_actions(3, np.arange(0, 8), np.arange(5))
    hi = rng.random_actions(3, np.arange(8, 16), np.arange(5))
    assert np.array_equal(full[:, :8], lo) and np.array_equal(full[:, 8:], hi)
    assert np.array_equal(rng.random_actions(3, [5], [2, 4])[:, 0], full[[2, 4], 5])
    assert (np.abs(full[..., :4]) <= 87).all() and (np.abs(full[..., 4:7]) <= 12).all()
    assert (full[..., 7] >= 0).all() and (full[..., 7] < 255).all()
    u = rng.uniform(0, np.arange(4096), np.arange(8), 8)
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    n, s = rng.prop_params(0, np.arange(9))
    assert list(n) == [2, 3, 4, 2, 3, 4, 2, 3, 4]
    assert ((s >= 0.015) & (s <= 0.016)).all() and np.allclose(s[..., 0], s[..., 2])


def test_placement_respects_workspace_and_clearance():
    ids = np.arange(512)
    n, s = rng.prop_params(1, ids)
    lo, hi = (0.35, -0.4, 0.43), (0.55, 0.4, 0.435)
    pose, ok = placement.sample_poses(1, ids, n, s, lo, hi)
    assert ok.all()
    for i in range(len(ids)):
        k = int(n[i])
        p = pose[i, :k, :3]
        assert (p >= np.array(lo) - 1e-12).all() and (p <= np.array(hi) + 1e-12).all()
        assert np.isnan(pose[i, k:]).all()
        assert np.allclose(np.linalg.norm(pose[i, :k, 3:], axis=1), 1.0)
        assert (pose[i, :k, 4:6] == 0).all() and (pose[i, :k, 6] >= 0).all()  # yaw = pi*U(0,1)
        for a in range(k):
            for b in range(a):
                d = np.linalg.norm(p[a] - p[b])
                assert d > placement.PROP_MARGIN + 2 * 0.015
    # sharding independence
    p2, _ = placement.sample_poses(1, ids[100:200], n[100:200], s[100:200], lo, hi)
    assert np.array_equal(np.nan_to_num(p2), np.nan_to_num(pose[100:200]))


def test_run_controller_tick_count_follows_fp64_clock():
    """while physics.data.time - start < duration (robot_arm.py:68-69), 5 steps per tick."""
    from mujoco_robot_environments_amd.models.robot_arm import RobotArm

    class P:  # minimal stand-in for the device-backed physics
        num_envs, timestep = 1, 0.001
        def osc_configure(self, **k): pass
        def gripper_set(self, c): pass
    r = RobotArm(P(), controller_params=config.default_config().robots.arm.controller_config.controller_params)
    assert r.control_steps == 5
    for dur in (2.0, 1.0):
        t, start, ticks = r.time, r.time, 0
        while t - start < dur:
            for _ in range(5):
                t += 0.001
            ticks += 1
        assert r.ticks_for(dur) == ticks and abs(ticks - dur / 0.005) <= 1


def test_prop_bboxes_from_a_segmentation_image_on_cpu():
    """get_bbox of the reference (tasks/rearrangement.py:254-268) vectorised over the batch: PASCAL
    VOC corners of the visible pixels of each cube id, -1 where the id is absent."""
    import torch
    from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, PROP_GEOM_ID0
    seg = torch.full((2, 48, 64), 1, dtype=torch.uint8)
    seg[0, 10:15, 20:31] = PROP_GEOM_ID0          # cube 0: rows 10..14, cols 20..30
    seg[0, 40:42, 3:5] = PROP_GEOM_ID0 + 2        # cube 2
    seg[1, 0:1, 63:64] = PROP_GEOM_ID0 + 3        # a single pixel in the corner
    boxes = BatchedRearrangementEnv.prop_bboxes(None, seg)
    assert boxes.shape == (2, 4, 4)
    assert boxes[0, 0].tolist() == [20, 10, 30, 14]
    assert boxes[0, 2].tolist() == [3, 40, 4, 41]
    assert boxes[1, 3].tolist() == [63, 0, 63, 0]
    assert (boxes[0, 1] == -1).all() and (boxes[0, 3] == -1).all() and (boxes[1, :3] == -1).all()


def test_tuned_gain_helper_only_touches_the_gains():
    from mujoco_robot_environments_amd.config import (colour_separator_task_config, apply_tuned_osc_gains,
                                                       TUNED_OSC_GAINS)
    a, b = colour_separator_task_config(), apply_tuned_osc_gains(colour_separator_task_config())
    ga = a.robots.arm.controller_config.controller_params
    gb = b.robots.arm.controller_config.controller_params
    assert ga.gains.position.kp == 350.0 and ga.gains.position.kd == 20.0      # osc.yaml:6-8 untouched by default
    for k, (kp, kd) in TUNED_OSC_GAINS.items():
        assert gb.gains[k].kp == kp and gb.gains[k].kd == kd
    assert list(ga.nullspace.joint_config) == list(gb.nullspace.joint_config)
    assert ga.convergence.position_threshold == gb.convergence.position_threshold


def test_robot_self_collision_pairs_are_opt_in():
    """spec.default_scene(robot_self_pairs=True): robot-robot pairs whose bodies are at least four arm links apart
    (gripper and fingers count as link 7) and the pads of opposite fingers -- 31 pairs on top of the 84; the emitted
    MJCF lists them as explicit <pair>s.  Nothing touches at the home pose; a gripper closing on nothing ends on its
    pads (driver 0.78 rad, short of the 0.8 limit); folding joint 6 to its lower limit stays free of the link-4 hull."""
    import os
    import sys
    import xml.etree.ElementTree as ET
    from mujoco_robot_environments_amd.model import spec as S
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import emit_mjcf
    from oracle import oracle as O
    scene = S.default_scene(dict(robot_self_pairs=True))
    A = MC.compile_scene(scene)
    assert int(A["npair"][0]) == 115 <= 128
    names = A["_names"]["geoms"]
    rr = [(names[a], names[b]) for a, b in A["pair_geom"]
          if not names[a].startswith(("table", "ground", "prop")) and not names[b].startswith(("table", "ground", "prop"))]
    assert len(rr) == 31 and ("right_pad1", "left_pad1") in rr and ("link5_hull", "link1_hull") in rr
    assert not any("link4" in a + b and ("pad" in a + b or "follower" in a + b) for a, b in rr)   # three links apart: dropped
    root = ET.fromstring(emit_mjcf.emit(scene, nprops=2))
    assert len(root.find("contact")) == 31
    om = O.Model(MC.to_blob(A))
    e = O.Env(om, 2, np.full((4, 3), 0.0155))
    e.set_solver("Newton")
    q = e.arr("qpos")
    q[:7] = A["home_qpos"]
    q[15:22] = [0.5, 0.2, 0.4155, 1, 0, 0, 0]
    q[22:29] = [0.45, -0.2, 0.4155, 1, 0, 0, 0]
    e.forward()

    def self_contacts():
        return sorted({(names[int(c[13])], names[int(c[14])]) for c in e.contacts() if c[12] < 0 and
                       not names[int(c[13])].startswith(("table", "prop")) and not names[int(c[14])].startswith(("table", "prop"))})
    assert self_contacts() == []
    e.arr("ctrl")[:7] = e.arr("qfrc_bias")[:7]
    e.arr("ctrl")[7] = 255.0
    e.step(600)
    assert self_contacts() == [("right_pad1", "left_pad1")] and 0.77 < q[7] < 0.8 and abs(q[7] - q[11]) < 1e-4
