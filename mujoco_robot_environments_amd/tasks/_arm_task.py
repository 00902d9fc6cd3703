"""Shared body of the reference's three small environments -- ``BaseEnv`` (tasks/base.py), ``PushEnv``
(tasks/push.py) and ``LasaDrawEnv`` (tasks/lasa_draw.py) -- batched over ``num_envs`` and run on the SAME
step kernels as ``RearrangementEnv``.

Their hot path is the same five-substep pattern (tasks/base.py:228-254, tasks/push.py:339-363,
tasks/lasa_draw.py:300-370): one OSC torque (or one joint-position command) held for five
``physics.step()`` calls.  ``step()`` itself only renders, as in the reference.

The kernels are compiled for one body topology (arm + 2F-85 + four cube slots).  A task without a gripper
(push, lasa) is embedded in it: ``model/spec.py:other_task_scene`` hangs an INERT copy of the gripper
(no mass, no geoms, no springs) below the attachment body, so the arm rows of the mass matrix, the bias
forces and every contact are those of the arm-only model; ``tests/test_other_tasks.py`` checks in the fp64
oracle that the embedded model and the task's own body tree give the same arm and block trajectories.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
from scipy.spatial.transform import Rotation as R

from ..config import Cfg
from ..model import compile as _compile
from ..model import spec as _spec
from ..models.robot_arm import RobotArm
from ..physics import BatchedPhysics
from .rearrangement import StepType, TimeStep, _Array


def _actuator_cfg(ac) -> Dict:
    """robots.arm.actuator_config (motor.yaml | position.yaml) -> the scene builder's actuator arguments."""
    if ac.type == "motor":
        lim = [float(ac[ac.joint_actuator_mapping[f"joint{i + 1}"]].ctrlrange.split()[1]) for i in range(7)]
        return dict(actuator="motor", motor_ctrlrange=lim)
    rows = []
    for i in range(7):
        j = ac[ac.joint_actuator_mapping[f"joint{i + 1}"]]
        cr = tuple(float(x) for x in str(j.ctrlrange).split())
        fr = tuple(float(x) for x in str(j.forcerange).split())
        bias = tuple(float(x) for x in str(j.biasprm).split())
        rows.append((cr, fr[1], float(j.gainprm), -bias[2]))
        assert bias[0] == 0.0 and bias[1] == -float(j.gainprm), "position actuator: biasprm = 0 -kp -kv"
    return dict(actuator="position", position_actuators=rows)


class BatchedArmTaskEnv:
    """num_envs independent instances of one of the small tasks, stepped in lockstep on one GPU."""

    TASK = "base"
    CAMERA = "overhead_camera"
    MOCAP_POS = (0.4, 0.0, 0.2)
    TARGET_OFFSET = 0.175
    HAS_GRIPPER = True

    def __init__(self, cfg: Cfg, num_envs: int = 1, viewer=None, device: int = 0, render: bool = False,
                 solver: str = "Newton", scene_cfg: Optional[dict] = None):
        self._cfg = cfg
        self.num_envs = int(num_envs)
        self.has_viewer = False  # no viewer on a headless GPU batch
        sc = dict(physics_dt=cfg.physics_dt, gravity=cfg.gravity, solver=solver,
                  home=cfg.robots.arm.default_configurations.home)
        sc.update(_actuator_cfg(cfg.robots.arm.actuator_config))
        sc.update(scene_cfg or {})
        self.position_actuators = sc["actuator"] == "position"
        self._scene = _spec.other_task_scene(self.TASK, sc)
        self._model = _compile.compile_scene(self._scene)
        self._physics = BatchedPhysics(self.num_envs, model=self._model, device=device)
        self.nprops = np.full(self.num_envs, 1 if self.TASK == "push" else 0, np.int32)
        self.prop_half_size = np.full((self.num_envs, 4, 3), 0.025 if self.TASK == "push" else 0.0155)
        self._physics.set_props(self.nprops, self.prop_half_size)
        # cameras: pose / fovy constants of config/arena/cameras/*.yaml
        self._cameras = {}
        self.camera_height = self.camera_width = None
        for cam in cfg.arena.cameras:
            q = np.asarray(cam.quat, np.float64)
            q = q / np.linalg.norm(q)
            self._cameras[f"{cam.name}/{cam.name}"] = dict(pos=np.asarray(cam.pos, np.float64), mat=_compile.q2m(q),
                                                           fovy=float(cam.fovy), height=int(cam.height), width=int(cam.width))
            if cam.name == self.CAMERA:
                self.camera_height, self.camera_width = int(cam.height), int(cam.width)
        if self.camera_height is None:
            raise ValueError(f"config has no camera named {self.CAMERA}")
        self.render_observations = bool(render)
        names = self._model["_names"]["geoms"]
        geom_rgb = np.full((len(names), 3), 0.25, np.float32)
        for g, n in enumerate(names):
            if n.startswith("table"):
                geom_rgb[g] = 1.0      # tasks/push.py:90, tasks/lasa_draw.py:93: white tables
            if n == "tool_cylinder":
                geom_rgb[g] = (0.02, 0.302, 0.4)
        if self.TASK == "push":      # green -> red gradient of the eight slabs (tasks/push.py:103-116)
            for k in range(8):
                t = k / 7.0
                geom_rgb[names.index(f"table_{k + 1}")] = (t, 1.0 - t, 0.0)
        prop_rgb = np.full((self.num_envs, 4, 3), 128, np.uint8)   # push block: rgba 0.5 0.5 0.5
        self._physics.set_render_colours(prop_rgb, geom_rgb)
        # the reference's mocap target body; R.as_quat() is (x, y, z, w) and is handed to MuJoCo's
        # (w, x, y, z) unconverted there: reproduced
        mq = R.from_euler("xyz", [180, 180, 0], degrees=True).as_quat()
        self.mocap_pos = np.tile(np.asarray(self.MOCAP_POS, np.float64), (self.num_envs, 1))
        self.mocap_quat = np.tile(mq / np.linalg.norm(mq), (self.num_envs, 1))
        self._robot: Optional[RobotArm] = None
        self.eef_home_pose = None

    # ------------------------------------------------------------------ misc
    def close(self) -> None:
        self._physics.close()

    @property
    def physics(self) -> BatchedPhysics:
        return self._physics

    @property
    def model(self) -> dict:
        return self._model

    @property
    def data(self) -> BatchedPhysics:
        return self._physics

    def render(self, rgb: bool = True, depth: bool = True, seg: bool = False, camera: Optional[str] = None):
        cam = self._cameras[camera or f"{self.CAMERA}/{self.CAMERA}"]
        return self._physics.render(cam["pos"], cam["mat"], cam["fovy"], cam["height"], cam["width"],
                                    rgb=rgb, depth=depth, seg=seg)

    def _compute_observation(self):
        h, w, n = self.camera_height, self.camera_width, self.num_envs
        if not self.render_observations:
            rgb = np.broadcast_to(np.zeros((1, 1, 1, 1), np.uint8), (n, h, w, 3))
            depth = np.broadcast_to(np.zeros((1, 1, 1), np.float32), (n, h, w))
        else:
            rgb, depth, _ = self.render()
        return {f"{self.CAMERA}/rgb": rgb, f"{self.CAMERA}/depth": depth}

    def observation_spec(self):
        h, w = self.camera_height, self.camera_width
        return {f"{self.CAMERA}/depth": _Array(shape=(h, w), dtype=np.float32),
                f"{self.CAMERA}/rgb": _Array(shape=(h, w, 3), dtype=np.float32)}

    def action_spec(self) -> Dict[str, _Array]:
        return {"pose": _Array(shape=(7,), dtype=np.float64),
                "pixel_coords": _Array(shape=(2,), dtype=np.int64),
                "gripper_rot": _Array(shape=(1,), dtype=np.float64)}

    # ----------------------------------------------------------------- reset
    def _initial_qpos(self, qpos: np.ndarray) -> None:
        """Task-specific initial state written into qpos [N, 43] (arm already at home)."""

    def reset(self, arm_configuration=None) -> TimeStep:
        """physics.reset(), arm to the home (or given) configuration, a new RobotArm
        (tasks/base.py:150-190, tasks/push.py:241-283, tasks/lasa_draw.py:207-244)."""
        self._physics.reset()
        qpos, qvel = self._physics.get_state()
        qpos, qvel = qpos.copy(), qvel.copy()
        if arm_configuration is not None:
            qpos[:, :7] = np.asarray(arm_configuration, np.float32)
        self._initial_qpos(qpos)
        self._physics.set_state(qpos, qvel)
        cp = self._cfg.robots.arm.controller_config.controller_params
        mm = self._cfg.robots.end_effector.controller_config.controller
        self._robot = RobotArm(self._physics, controller_params=cp, gripper_cfg=mm, gripper=self.HAS_GRIPPER,
                               strict=False)
        self.eef_home_pose = np.atleast_2d(self._robot.eef_pose).copy()
        return TimeStep(step_type=StepType.FIRST, reward=0.0, discount=0.0, observation=self._compute_observation())

    def step(self, action_dict=None) -> TimeStep:
        """tasks/base.py:192-203 and its copies: the observation, nothing else."""
        return TimeStep(step_type=StepType.MID, reward=0.0, discount=0.0, observation=self._compute_observation())

    # ------------------------------------------------------------- hot path
    def _osc_tick(self, position, quat, velocity) -> None:
        self._robot.arm_controller.set_target(position=position, quat=quat, velocity=velocity,
                                              angular_velocity=np.zeros(3))
        # compute_control_output() once, then 5 x (set_control, step): one control tick of the fused launch
        self._physics.run_controller(1, 5)
        for _ in range(5):
            self._robot.time += self._robot.timestep

    def interactive_tuning(self, mocap_pos=None, mocap_quat=None) -> None:
        """One pass of the tuning loop: the OSC target follows the mocap body (+ TARGET_OFFSET in z), the
        command is computed once and held for five physics steps."""
        if self.position_actuators:
            raise RuntimeError("interactive_tuning drives motors through the OSC torque law; "
                               "this env was built with position actuators")
        if mocap_pos is not None:
            self.mocap_pos[:] = np.asarray(mocap_pos, np.float64)
        if mocap_quat is not None:
            self.mocap_quat[:] = np.asarray(mocap_quat, np.float64)
        self._osc_tick(self.mocap_pos + np.array([0.0, 0.0, self.TARGET_OFFSET]), self.mocap_quat, np.zeros(3))
