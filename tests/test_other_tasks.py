"""The reference's other environments (tasks/base.py, tasks/push.py, tasks/lasa_draw.py) -- CPU side:
scene compilation into the kernels' topology, the embedding of an arm-only task checked in the fp64 oracle
against the task's own body tree, the position actuators, and the quirks of those files the mirror states."""
import numpy as np
import pytest

from mujoco_robot_environments_amd import config as cfgm
from mujoco_robot_environments_amd.model import compile as MC
from mujoco_robot_environments_amd.model import spec
from oracle import oracle as O


def _oracle(task, cfg=None, embed=True):
    A = MC.compile_scene(spec.other_task_scene(task, cfg, embed=embed))
    M = O.Model(MC.to_blob(A))
    e = O.Env(M, 1 if task == "push" else 0, np.full((4, 3), 0.025 if task == "push" else 0.0155))
    e.reset()
    e.arr("qpos")[:7] = A["home_qpos"]
    if task == "push":
        pa = int(A["body_qposadr"][A["prop_bodyid"][0]])
        e.arr("qpos")[pa:pa + 7] = [0.3, 0.0, 0.6, 0.0, 0.0, 0.0, 1.0]
    e.forward()
    return A, M, e


@pytest.mark.parametrize("task", ["base", "push", "lasa"])
def test_scenes_compile_into_the_kernel_topology(task, compiled_model):
    ref = compiled_model[0]
    A = MC.compile_scene(spec.other_task_scene(task))
    for k in ("nbody", "nv", "nq", "nu", "ngeom", "nsite", "neq", "nprop", "nM"):
        assert int(A[k][0]) == int(ref[k][0]), k
    assert int(A["npair"][0]) <= 128
    np.testing.assert_array_equal(A["dof_parentid"], ref["dof_parentid"])
    np.testing.assert_array_equal(A["geom_propid"][12:16], [0, 1, 2, 3])
    assert A["geom_type"][1] == 1 and A["geom_bodyid"][1] == 0
    if task != "base":   # the inert gripper: no mass below the arm's last link, armature only
        assert np.all(A["body_mass"][8:16] == 0.0) and np.all(A["dof_armature"][7:15] > 0)
        assert np.all(A["jnt_stiffness"][8:16] == 0.0) and np.all(A["dof_damping"][7:15] == 0.0)
        assert float(A["grip_gainprm"][0]) == 0.0
        # link 7 carries the tool cylinder: density 1000, radius 0.015, half height 0.05
        assert A["body_mass"][7] == pytest.approx(0.735522 + 1000 * np.pi * 0.015 ** 2 * 0.1, rel=1e-6)
    else:
        np.testing.assert_allclose(A["body_mass"][1:16], ref["body_mass"][1:16])


def test_push_embedding_equals_the_arm_only_model():
    """OSC drags the tool into the block for 1.5 s: the embedded model (inert gripper, 39 dofs) and the task's own
    body tree (13 dofs) give the same arm and block trajectory in the fp64 oracle."""
    traj = {}
    for embed in (True, False):
        A, M, e = _oracle("push", embed=embed)
        p = O.make_osc()
        pa = int(A["body_qposadr"][A["prop_bodyid"][0]])
        out = []
        for k in range(300):
            p.target_pos[:] = [0.2 + 0.0006 * k, 0.0, 0.53]
            p.target_quat[:] = [0, 0, 1, 0]
            e.run_controller(p, 0.0, 1, 5)
            q = e.arr("qpos")
            out.append(np.concatenate([q[:7], q[pa:pa + 7]]))
            if embed:
                assert np.abs(q[7:15]).max() < 1e-6   # the inert fingers stay put (float32-rounded anchors: 1e-8)
        traj[embed] = np.array(out)
        if embed:
            assert traj[embed][-1][7] > 0.33, "the block was pushed"
    d = np.abs(traj[True] - traj[False])
    assert d[:, :7].max() < 1e-8 and d[:, 7:].max() < 1e-7, (d[:, :7].max(), d[:, 7:].max())


def test_lasa_position_actuators_embedding_and_law():
    """Deployment config: kp (ctrl - q) - kv qvel clamped to the force range, through implicitfast's actuator
    velocity derivative, at physics_dt 0.01; embedded == arm-only to round-off."""
    traj = {}
    for embed in (True, False):
        A, M, e = _oracle("lasa", dict(actuator="position"), embed=embed)
        assert float(A["opt_timestep"][0]) == 0.01
        out = []
        for k in range(150):
            tgt = np.array(A["home_qpos"]) + 0.3 * np.sin(0.05 * k + np.arange(7))
            if 60 <= k < 70:
                tgt[1] += 1.0    # a jump: joint 2's 4500 N m / rad saturates the +-87 N m force range
            e.arr("ctrl")[:7] = tgt
            e.step(1)
            q, v = e.arr("qpos")[:7], e.arr("qvel")[:7]
            f = e.arr("actuator_force")[:7]
            if 61 <= k < 64:
                assert abs(f[1]) == 87.0
            e.step(4)
            out.append(e.arr("qpos")[:7].copy())
        traj[embed] = np.array(out)
        assert np.abs(traj[embed][59] - (np.array(A["home_qpos"]) + 0.3 * np.sin(0.05 * 59 + np.arange(7)))).max() < 0.05
    assert np.abs(traj[True] - traj[False]).max() < 1e-12


def test_position_actuator_force_identity():
    A, M, e = _oracle("lasa", dict(actuator="position"), embed=True)
    e.arr("qvel")[:7] = 0.01 * np.arange(1, 8)
    c = np.array(A["home_qpos"]) + 0.002 * np.arange(7)
    e.arr("ctrl")[:7] = c
    e.forward()
    kp = np.array([4500, 4500, 3500, 3500, 2000, 2000, 2000.0])
    want = kp * (c - e.arr("qpos")[:7]) - 0.1 * kp * e.arr("qvel")[:7]
    np.testing.assert_allclose(e.arr("actuator_force")[:7], want, rtol=1e-6, atol=1e-6)


def test_lasa_default_config_quirk_torque_control_at_10ms():
    """config/lasa.yaml runs the OSC torque law with a 50 ms hold (5 x physics_dt 0.01): kd_ori * 0.05 s = 5 > 2,
    the sampled velocity loop is unstable.  Holding the home pose diverges at 0.01 and is exact at 0.001."""
    worst = {}
    for dt in (0.01, 0.001):
        A, M, e = _oracle("lasa", dict(physics_dt=dt))
        s = int(A["eef_site"][0])
        p = O.make_osc()
        p.target_pos[:] = e.arr("site_xpos").reshape(-1, 3)[s]
        p.target_quat[:] = MC.m2q(e.arr("site_xmat").reshape(-1, 9)[s].reshape(3, 3))
        mx = 0.0
        for _ in range(100):
            e.run_controller(p, 0.0, 1, 5)
            mx = max(mx, float(np.abs(e.arr("qvel")[:7]).max()))
        worst[dt] = mx
    assert worst[0.001] < 1e-5 and worst[0.01] > 5.0, worst


def test_push_friction_quirk_and_the_forwarded_gradient():
    """Rectangle._build drops its friction argument (environment/props.py:226-260): by default every slab and the
    block keep (1, 0.005, 0.0001); forward_friction=True applies tasks/push.py:112-125's gradient.  A block
    sliding on a slab decelerates at (impedance x) mu g."""
    dec = {}
    for ff in (False, True):
        A = MC.compile_scene(spec.other_task_scene("push", dict(forward_friction=ff)))
        names = A["_names"]["geoms"]
        blk = names.index("prop_0")
        mus = {}
        for k in range(A["npair"][0]):
            g1, g2 = A["pair_geom"][k]
            if blk in (g1, g2) and names[g1 + g2 - blk].startswith("table_"):
                mus[names[g1 + g2 - blk]] = A["pair_friction"][k][0]
        if not ff:
            assert set(mus.values()) == {1.0}
        else:
            assert mus["table_0"] == 1.0 and mus["table_1"] == pytest.approx(0.4) and mus["table_8"] == pytest.approx(0.8)
            assert all(mus[f"table_{k}"] < mus[f"table_{k + 1}"] for k in range(1, 8))
        M = O.Model(MC.to_blob(A))
        for xc in (0.2, 0.9):
            e = O.Env(M, 1, np.full((4, 3), 0.025))
            e.reset()
            e.arr("qpos")[:7] = A["home_qpos"]
            e.arr("qpos")[15:22] = [xc, 0.3, 0.425, 1, 0, 0, 0]
            e.freeze_robot(True)
            e.forward()
            e.step(300)
            e.arr("qvel")[16] = -0.5
            e.forward()   # (velocity-dependent terms of the coming step2 at the new velocity)
            e.step(50)
            dec[(ff, xc)] = (e.arr("qvel")[16] + 0.5) / 0.05
    g = 9.8
    for key, mu in (((False, 0.2), 1.0), ((False, 0.9), 1.0), ((True, 0.2), 0.4), ((True, 0.9), 0.8)):
        assert 0.9 * mu * g < dec[key] < 1.05 * mu * g, (key, dec)   # (pyramidal cones, along a tangent axis: 0.98 .. 1.03)


def test_configs_of_the_other_tasks():
    l = cfgm.lasa_default_config()
    assert l.physics_dt == 0.01 and l.simulation_tuning_mode is True
    assert [c.name for c in l.arena.cameras] == ["main_camera"]   # config/arena/cameras/lasa.yaml (one live entry)
    assert "props" not in l.arena                                 # config/arena/lasa.yaml has no props group
    d = cfgm.lasa_deployment_config()
    assert d.robots.arm.actuator_config.type == "general"
    assert d.robots.arm.actuator_config.joint4.ctrlrange == "-3.0718 -0.0698"
    from mujoco_robot_environments_amd.tasks._arm_task import _actuator_cfg
    a = _actuator_cfg(d.robots.arm.actuator_config)
    assert a["actuator"] == "position" and a["position_actuators"][4] == ((-2.8973, 2.8973), 12.0, 2000.0, 200.0)
    assert _actuator_cfg(cfgm.push_default_config().robots.arm.actuator_config)["motor_ctrlrange"][4] == 12.0


@pytest.mark.parametrize("task", ["base", "push", "lasa"])
def test_emitted_mjcf_of_the_other_tasks(task):
    """tools/emit_mjcf.py --task: the task's OWN body tree (no inert gripper) with the pair filter of the compiled
    model -- what a box with `import mujoco` would step next to the oracle."""
    import os
    import sys
    import xml.etree.ElementTree as ET
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import emit_mjcf
    scene = spec.other_task_scene(task, dict(actuator="position" if task == "lasa" else "motor"), embed=False)
    A = MC.compile_scene(scene)
    n = 1 if task == "push" else 0
    root = ET.fromstring(emit_mjcf.emit(scene, nprops=n, prop_sizes=[[0.025] * 3] * 4))
    hinges = list(root.find("worldbody").iter("joint"))
    assert len(hinges) == (15 if task == "base" else 7) and len(list(root.find("worldbody").iter("freejoint"))) == n
    geoms = list(root.find("worldbody").iter("geom"))
    assert len(geoms) == int(A["ngeom"][0])
    assert float(root.find("option").get("timestep")) == (0.01 if task == "lasa" else 0.001)
    assert len(root.find("actuator")) == (8 if task == "base" else 7) and len(root.find("equality")) == (3 if task == "base" else 0)
    if task == "lasa":
        g = root.find("actuator")[3]
        assert g.tag == "general" and g.get("biasprm") == "0.0 -3500.0 -350.0" and g.get("forcerange") == "-87.0 87.0"
    if task != "base":   # the attachment body carries the cylinder's inertia explicitly
        att = [b for b in root.iter("body") if b.get("name") == "attachment"][0]
        assert float(att.find("inertial").get("mass")) == pytest.approx(1000 * np.pi * 0.015 ** 2 * 0.1, rel=1e-6)
    masks = [(int(g.get("contype")), int(g.get("conaffinity"))) for g in geoms]
    static = [g.get("name").startswith(("table", "ground")) for g in geoms]
    ncoll = sum(1 for a in range(len(masks)) for b in range(a + 1, len(masks))
                if ((masks[a][0] & masks[b][1]) or (masks[b][0] & masks[a][1])) and not (static[a] and static[b]))
    explicit = len(root.find("contact")) if root.find("contact") is not None else 0   # robot self-collision pairs
    assert ncoll + explicit == int(A["npair"][0]), (ncoll, explicit, int(A["npair"][0]))


def test_oracle_pyramidal_cone_rest_diamond_and_what_the_elliptic_stand_in_cost():
    """PushEnv / LasaDrawEnv set neither `cone` nor `impratio`: the reference's MuJoCo runs them with pyramidal cones
    (four one-sided edge rows n +- mu t_k per contact, R = 2 mu^2 R0 on each, diagApprox = tran (1 + mu^2)).  The
    oracle builds those rows; three known answers: (1) a resting block sinks exactly as deep as with an elliptic cone
    (MuJoCo scales the edges' R for that); (2) sliding friction is a diamond -- mu g along the contact frame's
    tangent axes, about mu g / sqrt(2) along their diagonal -- where an elliptic cone brakes both alike; (3) what
    rounds 1-2 paid for running these tasks with elliptic cones: 3 mm of block travel over an 11 cm push."""
    rest, dec = {}, {}
    for cone in ("elliptic", "pyramidal"):
        for solver in ("Newton", "PGS"):
            A = MC.compile_scene(spec.other_task_scene("push", dict(cone=cone, solver=solver)))
            assert int(A["opt_cone"][0]) == (0 if cone == "pyramidal" else 1)
            M = O.Model(MC.to_blob(A))
            for name, vel in (("axis", (-0.5, 0.0)), ("diag", (-0.5 / np.sqrt(2), -0.5 / np.sqrt(2)))):
                e = O.Env(M, 1, np.full((4, 3), 0.025))
                e.reset()
                e.arr("qpos")[:7] = A["home_qpos"]
                e.arr("qpos")[15:22] = [0.5, 0.3, 0.425, 1, 0, 0, 0]
                e.freeze_robot(True)
                e.forward()
                e.step(300)
                assert e.nefc - 7 - e.nl == (16 if cone == "pyramidal" else 12)   # four corner contacts
                rest[(cone, solver)] = e.arr("qpos")[17] - 0.425
                e.arr("qvel")[15:17] = vel
                e.forward()
                e.step(30)
                v = e.arr("qvel")[15:17]
                dec[(cone, solver, name)] = (0.5 - np.hypot(*v)) / 0.03 / 9.8
                assert abs(v[0] * vel[1] - v[1] * vel[0]) < 2e-2 * 0.5 * np.hypot(*v)   # braked along its motion
    for solver in ("Newton", "PGS"):
        assert abs(rest[("pyramidal", solver)] - rest[("elliptic", solver)]) < 5e-7, rest
        assert -1.2e-4 < rest[("pyramidal", solver)] < -1.0e-4
        assert 0.93 < dec[("pyramidal", solver, "axis")] < 1.02, dec
        assert 0.78 < dec[("pyramidal", solver, "diag")] < 0.88, dec
        assert abs(dec[("elliptic", solver, "diag")] / dec[("elliptic", solver, "axis")] - 1.0) < 0.08, dec
    # (3) the push of test_push_embedding_equals_the_arm_only_model under both cones
    traj = {}
    for cone in ("elliptic", "pyramidal"):
        A, M, e = _oracle("push", dict(cone=cone))
        p = O.make_osc()
        pa = int(A["body_qposadr"][A["prop_bodyid"][0]])
        out = []
        for k in range(300):
            p.target_pos[:] = [0.2 + 0.0006 * k, 0.0, 0.53]
            p.target_quat[:] = [0, 0, 1, 0]
            e.run_controller(p, 0.0, 1, 5)
            out.append(e.arr("qpos")[pa:pa + 3].copy())
        traj[cone] = np.array(out)
    gap = np.abs(traj["elliptic"] - traj["pyramidal"]).max(axis=0)
    assert traj["pyramidal"][-1][0] > 0.40 and 5e-4 < gap[0] < 6e-3, gap
