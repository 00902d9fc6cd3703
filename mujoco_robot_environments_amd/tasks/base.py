"""BaseEnv: the reference's arm + gripper debug environment (tasks/base.py) on the batched GPU step.

Scene (tasks/base.py:78-87): Panda + 2F-85 standing on the arena floor (robot base at the origin), no table,
no props.  ``interactive_tuning`` (:228-254) is the hot path: OSC target = mocap pose + (0, 0, 0.175), the arm
torque and the gripper command computed once and held for five physics steps."""
from __future__ import annotations

from typing import Optional

from ..config import Cfg, default_config
from ._arm_task import BatchedArmTaskEnv


class BatchedBaseEnv(BatchedArmTaskEnv):
    TASK = "base"
    CAMERA = "overhead_camera"
    MOCAP_POS = (0.4, 0.0, 0.2)   # tasks/base.py:95
    TARGET_OFFSET = 0.175         # tasks/base.py:239
    HAS_GRIPPER = True

    def __init__(self, cfg: Optional[Cfg] = None, num_envs: int = 1, **kw):
        super().__init__(cfg if cfg is not None else default_config(), num_envs=num_envs, **kw)


class BaseEnv(BatchedBaseEnv):
    """Batch of one with the reference's signature (tasks/base.py:48-52)."""

    def __init__(self, viewer=None, cfg: Optional[Cfg] = None, **kw):
        super().__init__(cfg=cfg, num_envs=1, viewer=viewer, **kw)

    def _compute_observation(self):
        o = super()._compute_observation()
        return {k: (v[0].cpu().numpy() if hasattr(v, "cpu") else v[0]) for k, v in o.items()}
