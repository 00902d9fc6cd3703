"""PushEnv: the reference's non-prehensile manipulation environment (tasks/push.py) on the batched GPU step.

Scene (tasks/push.py:83-221): nine table slabs, the arm WITHOUT a gripper but with a tool cylinder on its
attachment body, one free block (half size 0.025, mass 0.05) dropped from (0.3, 0, 0.6).
``interactive_tuning`` (:339-363) is the hot path: OSC target = mocap pose + (0, 0, 0.175), torque held for five
physics steps.  ``model/spec.py:other_task_scene`` documents the friction quirk this mirror preserves and the
``forward_friction`` switch."""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..config import Cfg, push_default_config
from ._arm_task import BatchedArmTaskEnv


class BatchedPushEnv(BatchedArmTaskEnv):
    TASK = "push"
    CAMERA = "overhead_camera"
    MOCAP_POS = (0.4, 0.0, 0.6)   # tasks/push.py:179
    TARGET_OFFSET = 0.175         # tasks/push.py:351
    HAS_GRIPPER = False
    BLOCK_POSE = (0.3, 0.0, 0.6, 0.0, 0.0, 0.0, 1.0)   # tasks/push.py:255-259: set_pose(position, quat)

    def __init__(self, cfg: Optional[Cfg] = None, num_envs: int = 1, forward_friction: bool = False, **kw):
        sc = dict(kw.pop("scene_cfg", None) or {})
        sc["forward_friction"] = bool(forward_friction)
        super().__init__(cfg if cfg is not None else push_default_config(), num_envs=num_envs, scene_cfg=sc, **kw)

    def _initial_qpos(self, qpos: np.ndarray) -> None:
        qpos[:, 15:22] = np.asarray(self.BLOCK_POSE, np.float32)

    def block_pose(self) -> np.ndarray:
        """[N, 7] position + quaternion (w, x, y, z) of the push block."""
        return self._physics.qpos()[:, 15:22].astype(np.float64)


class PushEnv(BatchedPushEnv):
    """Batch of one with the reference's signature (tasks/push.py:47-51)."""

    def __init__(self, viewer=None, cfg: Optional[Cfg] = None, **kw):
        super().__init__(cfg=cfg, num_envs=1, viewer=viewer, **kw)

    def _compute_observation(self):
        o = super()._compute_observation()
        return {k: (v[0].cpu().numpy() if hasattr(v, "cpu") else v[0]) for k, v in o.items()}
