"""RearrangementEnv on the MI355X batched physics step.

Drop-in shell of the reference's ``tasks/rearrangement.py`` (class RearrangementEnv,
:51-779): same constructor arguments, ``reset()`` / ``step(action_dict)`` 4-tuples and
observation shapes, ``pick`` / ``place`` scripted macros, ``sort_colours`` / ``prop_pick`` /
``prop_place`` demo helpers and camera maths -- with the inner loop
(``RobotArm.run_controller`` -> OSC + ``physics.step()``) executed by the HIP kernels for
``num_envs`` environments in lockstep.  ``BatchedRearrangementEnv`` carries a leading env
axis on every array; ``RearrangementEnv`` is the batch of one with reference shapes and
reference error behaviour (RuntimeError when a phase does not converge).

Observations: with ``render=True`` the batched overhead camera (csrc/mre_render.hip) renders
depth + RGB of every env and ``props_info`` boxes / ``pixel_2_world``
read the rendered segmentation / depth; with ``render=False`` (BASELINE configs[4]: "render
stubbed") they are zero images of the reference shapes and the pixel/world conversions use the
analytic pinhole model with the table plane for depth.

Solver: the reference sets timestep / gravity / nconmax / njmax only (tasks/rearrangement.py:77-80),
so its MuJoCo runs the default Newton solver -- the env's default here too (``solver="Newton"``);
``solver="PGS"`` selects the solver BASELINE.json's north_star prescribes for the bench.
"""
from __future__ import annotations

import collections
import enum
from typing import Dict, Optional

import numpy as np
from scipy.spatial.transform import Rotation as R

from .. import demo_logic, placement, rng
from ..config import Cfg, colour_separator_task_config, default_config  # noqa: F401
from ..model import compile as _compile
from ..model import spec as _spec
from ..models.robot_arm import RobotArm
from ..physics import BatchedPhysics

try:  # dm_env is what the reference returns; fall back to an equivalent tuple
    import dm_env
    from dm_env import specs as _specs
    TimeStep, StepType = dm_env.TimeStep, dm_env.StepType
    _Array = _specs.Array
except Exception:  # pragma: no cover - dm_env absent on the target image
    class StepType(enum.IntEnum):
        FIRST = 0
        MID = 1
        LAST = 2

    TimeStep = collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])

    class _Array:
        def __init__(self, shape, dtype, name=None):
            self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name

        def __repr__(self):
            return f"Array(shape={self.shape}, dtype={self.dtype}, name={self.name})"

DEFAULT_CONFIG = default_config()
TABLE_TOP_Z = 0.4
PICK_HEIGHT = 0.575     # tasks/rearrangement.py:362,405  ("hardcode for now :(")
PRE_PICK_HEIGHT = 0.9   # :364,408
PROP_GEOM_ID0 = 12      # geom ids of prop_0..3 in the compiled scene

PropsLabels = collections.namedtuple("PropsLabels", ["shape", "colour", "texture"])
# environment/props.py:13-20
COLOURS = {"red": (1.0, 0.0, 0.0), "green": (0.0, 1.0, 0.0), "blue": (0.0, 0.0, 1.0),
           "yellow": (1.0, 1.0, 0.0), "cyan": (0.0, 1.0, 1.0), "magenta": (1.0, 0.0, 1.0)}
COLOUR_NOISE = 0.1      # environment/props.py:274,290
OVERHEAD = "overhead_camera/overhead_camera"


def mat2quat(mat3x3) -> np.ndarray:
    """mujoco.mju_mat2Quat: (w, x, y, z)."""
    return _compile.m2q(np.asarray(mat3x3, np.float64).reshape(3, 3))


def home_quat() -> np.ndarray:
    """mju_mat2Quat(R.from_euler('xyz',[0,180,0])) (tasks/rearrangement.py:391-393)."""
    return mat2quat(R.from_euler("xyz", [0, 180, 0], degrees=True).as_matrix())


class BatchedRearrangementEnv:
    """num_envs independent RearrangementEnv instances stepped in lockstep on one GPU."""

    def __init__(self, cfg: Optional[Cfg] = None, num_envs: int = 1, viewer=None, device: int = 0,
                 seed: Optional[int] = None, env_id_offset: int = 0, env_ids=None, render: bool = False,
                 solver: str = "Newton"):
        self._cfg = cfg if cfg is not None else DEFAULT_CONFIG
        cfg = self._cfg
        self.num_envs = int(num_envs)
        self.has_viewer = False  # no viewer on a headless GPU batch (tasks/rearrangement.py:64-70)
        # global env ids key every random draw; explicit ids may repeat (envs sharing a scene)
        self.env_ids = (np.arange(env_id_offset, env_id_offset + self.num_envs) if env_ids is None
                        else np.asarray(env_ids, np.int64).reshape(self.num_envs))
        self._explicit_ids = env_ids is not None
        self.seed = int(cfg.task.initializers.seed if seed is None else seed)
        ac = cfg.robots.arm.actuator_config
        lim = [float(ac[ac.joint_actuator_mapping[f"joint{i + 1}"]].ctrlrange.split()[1]) for i in range(7)]
        scene = _spec.default_scene(dict(physics_dt=cfg.physics_dt, gravity=cfg.gravity, motor_ctrlrange=lim,
                                         home=cfg.robots.arm.default_configurations.home, solver=solver))
        self._model = _compile.compile_scene(scene)
        self._physics = BatchedPhysics(self.num_envs, model=self._model, device=device)
        # ---- props: count / size / colour per env (environment/props.py:583-639)
        pc = cfg.arena.props
        u = rng.uniform(self.seed ^ 0xC0105, self.env_ids, [0], 12)[0]
        if pc.min_objects == pc.max_objects:
            self.nprops = np.full(self.num_envs, pc.min_objects, np.int32)
        else:  # np.random.randint(min, max): max exclusive
            self.nprops = (pc.min_objects + np.floor(u[:, 0] * (pc.max_objects - pc.min_objects))).astype(np.int32)
        assert self.nprops.max() <= 4, "the compiled scene has 4 cube slots"
        size = pc.min_object_size + (pc.max_object_size - pc.min_object_size) * u[:, 1:5]
        self.prop_half_size = np.repeat(size[:, :, None], 3, axis=2)
        self.prop_colours = []
        for i in range(self.num_envs):
            cols = []
            for p in range(int(self.nprops[i])):
                cols.append(pc.colours[p] if p <= 1 else pc.colours[int(u[i, 5 + p] * len(pc.colours)) % len(pc.colours)])
            self.prop_colours.append(cols)
        self._physics.set_props(self.nprops, self.prop_half_size)
        # ---- cameras (config/arena/cameras/*.yaml); only the pose / fovy constants are needed
        self._cameras = {}
        for cam in cfg.arena.cameras:
            q = np.asarray(cam.quat, np.float64)
            q = q / np.linalg.norm(q)
            self._cameras[f"{cam.name}/{cam.name}"] = dict(pos=np.asarray(cam.pos, np.float64),
                                                           mat=_compile.q2m(q), fovy=float(cam.fovy))
            if cam.name == "overhead_camera":
                self.overhead_camera_height, self.overhead_camera_width = int(cam.height), int(cam.width)
        # ---- overhead camera images (mre_render): cube albedo = colour + noise (props.py:288-291),
        # table grey (tasks/rearrangement.py:91), robot hulls dark
        self.render_observations = bool(render)
        un = rng.uniform(self.seed ^ 0xC0FFEE, self.env_ids, [0], 12)[0].reshape(self.num_envs, 4, 3)
        self.prop_rgb = np.zeros((self.num_envs, 4, 3), np.uint8)
        self.prop_rgba = np.ones((self.num_envs, 4, 4))
        for i in range(self.num_envs):
            for p, c in enumerate(self.prop_colours[i]):
                rgbf = np.clip(np.asarray(COLOURS.get(c, (0.5, 0.5, 0.5))) + COLOUR_NOISE * (2 * un[i, p] - 1), 0.0, 1.0)
                self.prop_rgba[i, p, :3] = rgbf
                self.prop_rgb[i, p] = np.round(rgbf * 255)
        geom_rgb = np.full((int(self._model["ngeom"][0]), 3), 0.25, np.float32)
        geom_rgb[1] = 0.5
        self._physics.set_render_colours(self.prop_rgb, geom_rgb)
        self._reset_count = 0
        self._place_count = 0
        self._place_counts = np.zeros(self.num_envs, np.int64)  # prop_place calls per env (RNG key)
        self._zones = None  # [N, 4, 4] colour zone of every cube (set when the colours are drawn)
        # pose of the reference's mocap target body (tasks/rearrangement.py:130-140; simulation_tuning_mode)
        self.mocap_pos = np.tile(np.array([0.4, 0.0, 0.6]), (self.num_envs, 1))
        self.mocap_quat = np.tile(home_quat(), (self.num_envs, 1))
        self._robot: Optional[RobotArm] = None
        self.mode = None
        self.eef_home_pose = None
        self.last_converged = np.ones(self.num_envs, bool)
        self.placement_failed = np.zeros(self.num_envs, bool)   # reset(): PropPlacer found no pose for a cube
        self.not_settled = np.zeros(self.num_envs, bool)
        self.failed_phase = np.full(self.num_envs, "", dtype=object)

    # ------------------------------------------------------------------ misc
    def close(self) -> None:
        self._physics.close()

    @property
    def physics(self) -> BatchedPhysics:
        return self._physics

    @property
    def model(self) -> dict:
        """tasks/rearrangement.py:219-221: the compiled model (arrays of model/compile.py; there is no MjModel)."""
        return self._model

    @property
    def data(self) -> BatchedPhysics:
        """tasks/rearrangement.py:223-225: the batched state holder (qpos / qvel / ctrl / contacts accessors)."""
        return self._physics

    def _zeros_obs(self):
        h, w = self.overhead_camera_height, self.overhead_camera_width
        n = self.num_envs
        rgb = np.broadcast_to(np.zeros((1, 1, 1, 1), np.uint8), (n, h, w, 3))
        depth = np.broadcast_to(np.zeros((1, 1, 1), np.float32), (n, h, w))
        return {"overhead_camera/rgb": rgb, "overhead_camera/depth": depth}

    def render(self, rgb: bool = True, depth: bool = True, seg: bool = True, camera: str = OVERHEAD):
        """Overhead camera images of every env's current state as CUDA tensors (rgb uint8 [N,H,W,3],
        depth float32 [N,H,W], seg uint8 [N,H,W]: 12 + p = cube p, 1 table, 2..11 and 16..19 robot, 255 nothing):
        the batched form of the reference's renderer / depth_renderer / seg_renderer passes
        (tasks/rearrangement.py:254-280, 460-478)."""
        cam = self._cameras[camera]
        return self._physics.render(cam["pos"], cam["mat"], cam["fovy"], self.overhead_camera_height,
                                    self.overhead_camera_width, rgb=rgb, depth=depth, seg=seg)

    def prop_bboxes(self, seg=None) -> np.ndarray:
        """PASCAL-VOC boxes [N, 4 cubes, (xmin, ymin, xmax, ymax)] of the VISIBLE pixels of every cube
        in the segmentation image (get_bbox, tasks/rearrangement.py:254-268); -1 where a cube is not
        in view or not in use."""
        import torch
        if seg is None:
            seg = self.render(rgb=False, depth=False)[2]
        n, h, w = seg.shape
        out = torch.full((n, 4, 4), -1, dtype=torch.int64, device=seg.device)
        for p in range(4):
            m = seg == (PROP_GEOM_ID0 + p)
            cols, rows = m.any(dim=1), m.any(dim=2)
            vis = cols.any(dim=1)
            box = torch.stack([cols.int().argmax(dim=1), rows.int().argmax(dim=1),
                               w - 1 - cols.flip(1).int().argmax(dim=1), h - 1 - rows.flip(1).int().argmax(dim=1)], dim=1)
            out[:, p] = torch.where(vis[:, None], box, out[:, p])
        return out.cpu().numpy()

    def _compute_observation(self):
        if not self.render_observations:
            return self._zeros_obs()
        rgb, depth, _ = self.render(seg=False)
        return {"overhead_camera/rgb": rgb, "overhead_camera/depth": depth}

    def observation_spec(self):
        h, w = self.overhead_camera_height, self.overhead_camera_width
        return {"overhead_camera/depth": _Array(shape=(h, w), dtype=np.float32),
                "overhead_camera/rgb": _Array(shape=(h, w, 3), dtype=np.float32)}  # (sic: :448; images are uint8)

    def action_spec(self) -> Dict[str, _Array]:
        return {"pose": _Array(shape=(7,), dtype=np.float64),
                "pixel_coords": _Array(shape=(2,), dtype=np.int64),
                "gripper_rot": _Array(shape=(1,), dtype=np.float64)}

    # ----------------------------------------------------------------- reset
    def reset(self) -> TimeStep:
        """Physics.reset, arm to home, PropPlacer (sample + settle), new RobotArm
        (tasks/rearrangement.py:297-337)."""
        ws = self._cfg.task.initializers.workspace
        self._physics.reset()
        if self._explicit_ids:
            self._physics.set_env_ids(self.env_ids)
        else:
            self._physics.set_env_id_offset(int(self.env_ids[0]))
        # PropPlacer: contact-based rejection sampling, then settle with the robot frozen; every env
        # stops by itself (>= 0.3 s, <= 2 s, max|qvel| < 1e-3 and max|qacc| < 1e-2) inside the kernel
        self._physics.place_props(self.seed + 104729 * self._reset_count, ws.min_pose, ws.max_pose,
                                  settle_steps=300)
        self._reset_count += 1
        # mre_place_props flags the envs whose rejection sampling ran out of attempts (MRE_ST_PLACEMENT_FAILED) or whose
        # cubes never came to rest (MRE_ST_NOT_SETTLED) and returns MRE_OK.  The reference raises
        # RuntimeError(_REJECTION_SAMPLING_FAILED) out of reset() (prop_initializer.py:230-233) and the data-generation loop
        # drops the episode: a batch of one raises likewise, a batch carries the mask so that sort_colours() and
        # BatchedEpisodeLogger leave those envs out.
        st = self._physics.status()
        self.placement_failed = (st & 16) != 0
        self.not_settled = (st & 8) != 0
        if self.num_envs == 1 and self.placement_failed[0]:
            raise RuntimeError("Failed to find a non-colliding pose for some props")
        steps = int(np.abs(self._physics.settle_steps()).max())
        cp = self._cfg.robots.arm.controller_config.controller_params
        mm = self._cfg.robots.end_effector.controller_config.controller
        self._robot = RobotArm(self._physics, controller_params=cp, gripper_cfg=mm)
        self._robot.time = steps * self._physics.timestep
        pose0 = np.atleast_2d(self._robot.eef_pose).copy()
        pose0[:, 0] -= 0.1  # reference shifts x although the comment says "up" (App. D.5)
        self.eef_home_pose = pose0
        self.mode = "pick"
        self.last_converged[:] = True
        self.failed_phase[:] = ""
        return TimeStep(step_type=StepType.FIRST, reward=0.0, discount=0.0,
                        observation=self._compute_observation())

    # ------------------------------------------------------------------ step
    def step(self, action_dict) -> TimeStep:
        observation = self._compute_observation()  # rendered BEFORE acting (App. D.1)
        if self.mode == "pick":
            self.pick(action_dict["pose"])
            self.mode = "place"
        else:
            self.place(action_dict["pose"])
            self.mode = "pick"
        return TimeStep(step_type=StepType.MID, reward=0.0, discount=0.0, observation=observation)

    def _phase(self, name: str, duration: float):
        conv = np.atleast_1d(self._robot.run_controller(duration))
        newly = (~conv) & (self.failed_phase == "")
        self.failed_phase[newly] = name
        self.last_converged &= conv
        return conv

    def interactive_tuning(self, mocap_pos=None, mocap_quat=None):
        """One tick of the reference's tuning loop (tasks/rearrangement.py:753-779): the OSC target follows
        the mocap body (position + [0, 0, 0.175], its quaternion, zero velocities), the arm and gripper commands
        are computed once and held for 5 physics steps.  There is no viewer to drag a mocap body here, so its
        pose is state of the env (``mocap_pos`` [N, 3] / ``mocap_quat`` [N, 4], initialised like the reference's
        body at (0.4, 0, 0.6), gripper down) that the caller may move, per env, through the arguments."""
        if mocap_pos is not None:
            self.mocap_pos[:] = np.asarray(mocap_pos, np.float64)
        if mocap_quat is not None:
            self.mocap_quat[:] = np.asarray(mocap_quat, np.float64)
        self._robot.arm_controller.set_target(position=self.mocap_pos + np.array([0.0, 0.0, 0.175]),
                                              quat=self.mocap_quat, velocity=np.zeros(3),
                                              angular_velocity=np.zeros(3))
        # compute_control_output() + 5 x (set_control, step) = one control tick of the fused launch
        self._physics.run_controller(1, self._robot.control_steps)
        for _ in range(self._robot.control_steps):
            self._robot.time += self._robot.timestep

    def time_limit_exceeded(self) -> bool:
        """tasks/rearrangement.py:216-217 (the reference reads a module-level ``cfg.time_limit`` that its config
        tree does not define; absent means no limit)."""
        return self._robot.time >= float(self._cfg.get("time_limit", float("inf")))

    def _pose2d(self, pose):
        pose = np.asarray(pose)
        return pose.reshape(self.num_envs, 7) if pose.ndim == 1 else pose

    def pick(self, pose):
        """Scripted pick (tasks/rearrangement.py:358-399): pre-pick 2 s, descend 2 s,
        close 1 s, lift 2 s, home 2 s."""
        pose[..., 2] = PICK_HEIGHT  # in-place, like the reference
        p = self._pose2d(pose).astype(np.float64)
        pre = p.copy()
        pre[:, 2] = PRE_PICK_HEIGHT
        c = self._robot.arm_controller
        c.set_target(position=pre[:, :3], velocity=np.zeros(3), quat=pre[:, 3:], angular_velocity=np.zeros(3))
        self._phase("Failed to move arm to pre pick position", 2.0)
        c.set_target(position=p[:, :3])
        self._phase("Failed to move arm to pick position", 2.0)
        self._robot.end_effector_controller.status = "max"
        self._phase("Failed to close gripper", 1.0)
        c.set_target(position=pre[:, :3])
        self._phase("Failed to move arm to pre grasp position", 2.0)
        c.set_target(position=self.eef_home_pose, quat=home_quat())
        self._phase("Failed to move arm to home position", 2.0)

    def place(self, pose):
        """Scripted place (tasks/rearrangement.py:401-440)."""
        pose[..., 2] = PICK_HEIGHT
        p = self._pose2d(pose).astype(np.float64)
        pre = p.copy()
        pre[:, 2] = PRE_PICK_HEIGHT
        c = self._robot.arm_controller
        c.set_target(position=pre[:, :3], quat=pre[:, 3:])
        self._phase("Failed to move arm to pre place position", 2.0)
        c.set_target(position=p[:, :3])
        self._phase("Failed to move arm to place position", 2.0)
        self._robot.end_effector_controller.status = "min"
        self._phase("Failed to open gripper", 1.0)
        c.set_target(position=pre[:, :3])
        self._phase("Failed to move arm to pre place position", 2.0)
        c.set_target(position=self.eef_home_pose, quat=home_quat())
        self._phase("Failed to move arm to home position", 2.0)

    # ---------------------------------------------------------------- camera
    def _get_camera_intrinsics(self, camera_name, h, w, inverse=False):
        fov = self._cameras[camera_name]["fovy"]
        f = (1.0 / np.tan(np.deg2rad(fov) / 2)) * self.overhead_camera_height / 2.0
        return np.array([[-f, 0, (w - 1) / 2], [0, f, (h - 1) / 2], [0, 0, 1]])

    def _get_camera_extrinsics(self, camera_name):
        cam = self._cameras[camera_name]
        ext = np.eye(4)
        ext[:3, :3] = cam["mat"].T
        ext[:3, 3] = -cam["mat"].T @ cam["pos"]
        return ext

    def world_2_pixel(self, camera_name, coords):
        """tasks/rearrangement.py:533-548 (coords [3] or [N,3])."""
        K = self._get_camera_intrinsics(camera_name, self.overhead_camera_height, self.overhead_camera_width)
        E = self._get_camera_extrinsics(camera_name)
        c = np.atleast_2d(np.asarray(coords, np.float64))
        cam = (E @ np.concatenate([c, np.ones((len(c), 1))], axis=1).T).T
        cam = cam[:, :3] / cam[:, 3:4]
        img = (K @ cam.T).T
        img = img[:, :2] / img[:, 2:3]
        out = np.round(img).astype(np.int32)
        return out[0] if np.asarray(coords).ndim == 1 else out

    def pixel_2_world(self, camera_name, coords, env: int = 0):
        """tasks/rearrangement.py:505-531: the pixel's rendered depth pushed back through the pinhole
        model (env selects the image of a batch).  Without rendering the depth is that of the table
        plane: the pixel ray intersected with z = table top."""
        K = self._get_camera_intrinsics(camera_name, self.overhead_camera_height, self.overhead_camera_width)
        E = self._get_camera_extrinsics(camera_name)
        coords = np.asarray(coords, np.float64)
        ray_c = np.linalg.inv(K) @ np.concatenate([coords, np.ones(1)])
        cam = self._cameras[camera_name]
        if self.render_observations:
            rc = np.round(coords).astype(np.int32)  # coords_rounded (:509)
            mask = np.zeros(self.num_envs, np.uint8)
            mask[env] = 1
            _, dimg, _ = self._physics.render(cam["pos"], cam["mat"], cam["fovy"], self.overhead_camera_height,
                                              self.overhead_camera_width, rgb=False, seg=False, mask=mask)
            depth = float(dimg[env, rc[1], rc[0]])
        else:
            d_w = cam["mat"] @ (-ray_c)  # camera looks along -z
            depth = (TABLE_TOP_Z - cam["pos"][2]) / d_w[2]
        camc = np.concatenate([ray_c * (-depth), np.ones(1)])
        w = np.linalg.inv(E) @ camc
        return w[:3] / w[3]

    def get_camera_params(self, camera_name):
        return {"intrinsics": self._get_camera_intrinsics(camera_name, self.overhead_camera_height, self.overhead_camera_width),
                "extrinsics": self._get_camera_extrinsics(camera_name)}

    def get_camera_metadata(self):
        K = self._get_camera_intrinsics("overhead_camera/overhead_camera", self.overhead_camera_height, self.overhead_camera_width)
        E = self._get_camera_extrinsics("overhead_camera/overhead_camera")
        quat = R.from_matrix(E[:3, :3]).as_quat()
        # NB the reference reads the translation from row 3 of the 4x4 (always 0,0,0): kept (:563-565)
        return {"intrinsics": {"fx": K[0, 0], "fy": K[1, 1], "cx": K[0, 2], "cy": K[1, 2]},
                "extrinsics": {"x": E[3, 0], "y": E[3, 1], "z": E[3, 2],
                               "qx": quat[0], "qy": quat[1], "qz": quat[2], "qw": quat[3]}}

    # ------------------------------------------------------------ demo logic
    def props_info_env(self, i: int, prop_pose: Optional[np.ndarray] = None, bboxes: Optional[np.ndarray] = None) -> dict:
        """props_info of env i (tasks/rearrangement.py:227-295); keys are geom ids.  With rendering
        enabled the bounding boxes come from the segmentation image like the reference's (visible
        pixels, empty array when out of view); otherwise from the projected cube corners."""
        if prop_pose is None:
            prop_pose = self._physics.sites()[2]
        if bboxes is None and self.render_observations:
            bboxes = self.prop_bboxes()
        info = {}
        for p in range(int(self.nprops[i])):
            pos = prop_pose[i, p, :3].astype(np.float64)
            qw = prop_pose[i, p, 3:7].astype(np.float64)
            mat = _compile.q2m(qw / np.linalg.norm(qw))
            half = self.prop_half_size[i, p]
            corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * half
            px = self.world_2_pixel("overhead_camera/overhead_camera", pos + corners @ mat.T)
            bbox = np.array([px[:, 0].min(), px[:, 1].min(), px[:, 0].max(), px[:, 1].max()])
            if bboxes is not None:
                bbox = bboxes[i, p] if bboxes[i, p, 0] >= 0 else np.array([])
            info[PROP_GEOM_ID0 + p] = {
                "prop_name": f"prop_{p}", "position": pos,
                "orientation": R.from_matrix(mat).as_quat(),  # scipy (x,y,z,w) like the reference
                "rgba": self.prop_rgba[i, p].copy(), "bbox": bbox,
                "labels": PropsLabels("cube", self.prop_colours[i][p], "plain")}
        return info

    def prop_pick_env(self, i: int, prop_id: int, info: Optional[dict] = None) -> np.ndarray:
        """tasks/rearrangement.py:579-595."""
        info = info or self.props_info_env(i)
        obj_pose, obj_quat = info[prop_id]["position"], info[prop_id]["orientation"]
        m = R.from_quat(obj_quat).as_matrix()
        rz = abs(np.rad2deg(np.arctan2(m[1, 0], m[0, 0])))
        rz = min([rz, rz - 90])
        grasp = mat2quat(R.from_euler("xyz", [0, 180, rz], degrees=True).as_matrix())
        return np.concatenate([obj_pose, grasp])

    def prop_place_env(self, i: int, prop_id: int, min_pose=None, max_pose=None, info: Optional[dict] = None):
        """Collision-free place pose (tasks/rearrangement.py:597-665): uniform samples in the bounds,
        the cube moved there and physics.forward() evaluated on the device (mre_prop_place), rejected
        while a contact with a geom other than the table has dist <= 0.05."""
        ws = self._cfg.task.initializers.workspace
        lo = np.asarray(ws.min_pose if min_pose is None else min_pose, np.float64)
        hi = np.asarray(ws.max_pose if max_pose is None else max_pose, np.float64)
        prop = np.full(self.num_envs, -1, np.int32)
        prop[i] = prop_id - PROP_GEOM_ID0
        bounds = np.zeros((self.num_envs, 6))
        bounds[i, :3], bounds[i, 3:] = lo, hi
        pose, att = self._physics.prop_place(self.seed + 1, prop, bounds, self._place_counts * demo_logic.MAX_PLACE_ATTEMPTS,
                                             demo_logic.MAX_PLACE_ATTEMPTS, demo_logic.PLACE_CLEARANCE)
        if att[i] <= 0:
            raise Exception("Failed to find collision free place pose.")
        self._place_counts[i] += 1
        return pose[i]

    def sort_colours_env(self, i: int, info: Optional[dict] = None):
        """tasks/rearrangement.py:700-751 for env i."""
        info = info or self.props_info_env(i)
        task = self._cfg.task
        for prop_id, a in info.items():
            tl = task.target_locations[task.colour_target_map[a["labels"].colour]]
            lo = np.array([tl["location"][0] - tl["size"][0] / 2, tl["location"][1] - tl["size"][1] / 2, 0.4])
            hi = np.array([tl["location"][0] + tl["size"][0] / 2, tl["location"][1] + tl["size"][1] / 2, 0.4])
            x, y, _ = a["position"]
            if not (lo[0] <= x <= hi[0] and lo[1] <= y <= hi[1]):
                return True, self.prop_pick_env(i, prop_id, info), self.prop_place_env(i, prop_id, lo, hi, info)
        return False, None, None

    def sort_colours(self, peek: bool = False):
        """Batched sort_colours on the device (mre_sort_colours: selection, prop_pick and the
        prop_place rejection loop of tasks/rearrangement.py:700-751 run per env in one launch pair):
        (in_progress[N], pick_pose[N,7], place_pose[N,7]); envs with nothing left to do (or whose place
        sampling failed, recorded in ``failed_phase``) get their home pose as a no-op target."""
        if self._zones is None:
            lo, hi = demo_logic.target_bounds(self._cfg.task, self.prop_colours, 4)
            self._zones = np.concatenate([lo[..., :2], hi[..., :2]], axis=2)
        which, pick, place, att = self._physics.sort_colours(self.seed, self._place_counts, self._zones,
                                                             demo_logic.MAX_PLACE_ATTEMPTS, demo_logic.PLACE_CLEARANCE)
        prog = (which >= 0) & (att > 0) & ~self.placement_failed
        failed = (which >= 0) & (att < 0) & ~self.placement_failed
        if not peek:   # (peek: only asks which envs still have a cube to move; consumes no place draws)
            self._place_counts[prog] += 1
            self.failed_phase[failed] = "Failed to find collision free place pose."
        idle = ~prog
        home = np.atleast_2d(self.eef_home_pose)
        pick[idle, :3] = place[idle, :3] = home[idle] if len(home) == self.num_envs else home[0]
        pick[idle, 3:] = place[idle, 3:] = home_quat()
        return prog, pick, place

    def random_pick_and_place(self):
        """tasks/rearrangement.py:667-698 (always the first prop), batched."""
        poses = self._physics.sites()[2]
        ws = self._cfg.task.initializers.workspace
        pick = np.zeros((self.num_envs, 7))
        place = np.zeros((self.num_envs, 7))
        u = rng.uniform(self.seed + 2, self.env_ids, [self._place_count], 3)[0]
        self._place_count += 1
        for i in range(self.num_envs):
            info = self.props_info_env(i, poses)
            first = list(info.keys())[0]
            m = R.from_quat(info[first]["orientation"]).as_matrix()
            rz = np.rad2deg(np.arctan2(m[1, 0], m[0, 0]))
            grasp = mat2quat(R.from_euler("xyz", [0, 180, rz], degrees=True).as_matrix())
            pick[i] = np.concatenate([info[first]["position"], grasp])
            place[i] = np.concatenate([np.asarray(ws.min_pose) + (np.asarray(ws.max_pose) - np.asarray(ws.min_pose)) * u[i], grasp])
        return pick, place


class RearrangementEnv(BatchedRearrangementEnv):
    """Batch of one with the reference's shapes and error behaviour
    (signature: tasks/rearrangement.py:54-58)."""

    def __init__(self, viewer=None, cfg: Optional[Cfg] = None, device: int = 0, render: bool = True,
                 solver: str = "Newton"):
        super().__init__(cfg=cfg, num_envs=1, viewer=viewer, device=device, render=render, solver=solver)

    def _compute_observation(self):
        o = super()._compute_observation()
        return {k: (v[0].cpu().numpy() if hasattr(v, "cpu") else v[0]) for k, v in o.items()}

    def _phase(self, name, duration):
        conv = super()._phase(name, duration)
        if not conv[0]:  # tasks/rearrangement.py:371-399,413-440
            raise RuntimeError(name)
        return conv

    @property
    def props_info(self) -> dict:
        return self.props_info_env(0)

    def prop_pick(self, prop_id):
        return self.prop_pick_env(0, prop_id)

    def prop_place(self, prop_id, min_pose=None, max_pose=None):
        return self.prop_place_env(0, prop_id, min_pose, max_pose)

    def sort_colours(self):
        return self.sort_colours_env(0)

    def random_pick_and_place(self):
        a, b = super().random_pick_and_place()
        return a[0], b[0]

    def reset(self):
        ts = super().reset()
        self.eef_home_pose = self.eef_home_pose[0]
        return ts

    def pick(self, pose):
        self.eef_home_pose = np.atleast_2d(self.eef_home_pose)
        try:
            super().pick(pose)
        finally:
            self.eef_home_pose = self.eef_home_pose[0]

    def place(self, pose):
        self.eef_home_pose = np.atleast_2d(self.eef_home_pose)
        try:
            super().place(pose)
        finally:
            self.eef_home_pose = self.eef_home_pose[0]
