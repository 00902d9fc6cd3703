"""LasaDrawEnv: the reference's LASA drawing environment (tasks/lasa_draw.py) on the batched GPU step.

Scene (tasks/lasa_draw.py:83-128): one table, the arm WITHOUT a gripper but with the tool cylinder, no props;
config/lasa.yaml sets physics_dt = 0.01.  Three hot paths, all "one command, five physics steps":

* ``interactive_tuning`` (:300-324) -- OSC target = mocap pose + (0, 0, 0.1);
* ``move_to_draw_target(position, velocity)`` (:326-349) -- OSC position AND velocity target, returns the arm's
  (qpos, qvel, ctrl);
* ``move_to_joint_position_target(q)`` (:351-370) -- the deployment config's position actuators
  (config/robots/arm/actuator_config/position.yaml) driven directly; its xfrc_applied perturbation has scale 0.

Quirk of the reference's default config, reproduced and measured in the fp64 oracle
(tests/test_other_tasks.py): with physics_dt = 0.01 an OSC torque is held for 50 ms, and kd_ori * 0.05 s = 5 > 2
makes the sampled velocity loop unstable -- the torque-controlled variants only behave with
``physics_dt <= 0.002``; the position-actuator variant is stable at 0.01."""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..config import Cfg, lasa_default_config, lasa_deployment_config  # noqa: F401
from ._arm_task import BatchedArmTaskEnv


class BatchedLasaDrawEnv(BatchedArmTaskEnv):
    TASK = "lasa"
    CAMERA = "main_camera"
    MOCAP_POS = (0.6, -0.25, 0.55)   # tasks/lasa_draw.py:136
    TARGET_OFFSET = 0.1              # tasks/lasa_draw.py:312
    HAS_GRIPPER = False

    def __init__(self, cfg: Optional[Cfg] = None, num_envs: int = 1, **kw):
        super().__init__(cfg if cfg is not None else lasa_default_config(), num_envs=num_envs, **kw)

    def move_to_draw_target(self, target_position, target_velocity):
        """tasks/lasa_draw.py:326-349; returns (qpos [N, 7], qvel [N, 7], ctrl [N, 7])."""
        pos = np.asarray(target_position, np.float64) + np.array([0.0, 0.0, self.TARGET_OFFSET])
        self._osc_tick(pos, self.mocap_quat, np.asarray(target_velocity, np.float64))
        qpos, qvel = self._physics.get_state()
        tau = self._physics.ctrl()[:, :7]
        return qpos[:, :7].astype(np.float64), qvel[:, :7].astype(np.float64), tau.astype(np.float64)

    def move_to_joint_position_target(self, target_position):
        """tasks/lasa_draw.py:351-370: ctrl = joint position targets held for five physics steps."""
        if not self.position_actuators:
            raise RuntimeError("move_to_joint_position_target needs the deployment config "
                               "(robots/arm/actuator_config=position)")
        c = np.zeros((self.num_envs, 8), np.float32)
        c[:, :7] = np.asarray(target_position, np.float32)
        self._physics.set_control(c)
        self._physics.step(5)
        for _ in range(5):
            self._robot.time += self._robot.timestep
        return self._physics.qpos()[:, :7].astype(np.float64)


class LasaDrawEnv(BatchedLasaDrawEnv):
    """Batch of one with the reference's signature (tasks/lasa_draw.py:62-66)."""

    def __init__(self, viewer=None, cfg: Optional[Cfg] = None, **kw):
        super().__init__(cfg=cfg, num_envs=1, viewer=viewer, **kw)

    def _compute_observation(self):
        o = super()._compute_observation()
        return {k: (v[0].cpu().numpy() if hasattr(v, "cpu") else v[0]) for k, v in o.items()}

    def move_to_draw_target(self, target_position, target_velocity):
        q, v, c = super().move_to_draw_target(target_position, target_velocity)
        return q[0], v[0], c[0]

    def move_to_joint_position_target(self, target_position):
        return super().move_to_joint_position_target(target_position)[0]
