"""Operational-space controller: host mirror of ``mujoco_controllers.osc.OSC``
(third-party, absent; API from call sites models/robot_arm.py:37,71,83,
tasks/rearrangement.py:365-375 and automated_controller_tuning/...:56-65,164-168).
The law itself runs inside the HIP step kernel (csrc/mre_osc.h); this class only holds
targets / gains and forwards them through the C ABI."""
from __future__ import annotations

import numpy as np


class OSC:
    def __init__(self, physics, arm=None, controller_config=None):
        self.physics = physics
        self.arm = arm
        n = physics.num_envs
        self.eef_target_position = np.zeros((n, 3))
        self.eef_target_quat = np.tile([1.0, 0, 0, 0], (n, 1))
        self.eef_target_velocity = np.zeros((n, 3))
        self.eef_target_angular_velocity = np.zeros((n, 3))
        self.controller_gains = dict(position=dict(kp=350.0, kd=20.0), orientation=dict(kp=500.0, kd=100.0),
                                     nullspace=dict(kp=200.0, kd=30.0))
        self.nullspace_config = np.array([0, -0.785, 0, -2.356, 0, 1.571, 0.785])
        self.position_threshold, self.orientation_threshold = 5e-3, 68e-3
        if controller_config is not None:
            g = controller_config["gains"]
            self.controller_gains = {k: dict(kp=float(g[k]["kp"]), kd=float(g[k]["kd"]))
                                     for k in ("position", "orientation", "nullspace")}
            self.nullspace_config = np.asarray(controller_config["nullspace"]["joint_config"], float)
            self.position_threshold = float(controller_config["convergence"]["position_threshold"])
            self.orientation_threshold = float(controller_config["convergence"]["orientation_threshold"])
        self._last_converged = np.zeros(n, bool)
        self._push_config()

    def _push_config(self):
        g = self.controller_gains
        self.physics.osc_configure(
            gains=[g["position"]["kp"], g["position"]["kd"], g["orientation"]["kp"], g["orientation"]["kd"],
                   g["nullspace"]["kp"], g["nullspace"]["kd"]],
            null_q=self.nullspace_config, thresholds=[self.position_threshold, self.orientation_threshold])

    def set_gains_per_env(self, gains):
        """A population of gain sets evaluated as one batch (one candidate per env): gains [N, 6] =
        kp, kd of position / orientation / nullspace -- the batched form of assigning
        ``controller_gains`` per candidate in automated_controller_tuning/
        rearrangement_controller_tuning.py:164-168.  ``_push_config`` returns to the shared set."""
        self.physics.osc_configure_env(gains=np.asarray(gains, np.float32))

    def _b(self, x, w):
        x = np.asarray(x, np.float64)
        if x.ndim == 1:
            x = np.tile(x, (self.physics.num_envs, 1))
        assert x.shape == (self.physics.num_envs, w), x.shape
        return x

    def set_target(self, position=None, velocity=None, quat=None, angular_velocity=None, mask=None):
        """Any argument left None keeps its previous value (tasks/rearrangement.py:375)."""
        sel = slice(None) if mask is None else np.asarray(mask, bool)
        if position is not None:
            self.eef_target_position[sel] = self._b(position, 3)[sel]
        if velocity is not None:
            self.eef_target_velocity[sel] = self._b(velocity, 3)[sel]
        if quat is not None:
            self.eef_target_quat[sel] = self._b(quat, 4)[sel]
        if angular_velocity is not None:
            self.eef_target_angular_velocity[sel] = self._b(angular_velocity, 3)[sel]
        self.physics.osc_set_target(position=self.eef_target_position, quat=self.eef_target_quat,
                                    velocity=self.eef_target_velocity,
                                    angular_velocity=self.eef_target_angular_velocity)

    def compute_control_output(self):
        """Arm torques [N, 7] (or [7] for a batch of one) of the law on the current state
        (models/robot_arm.py:71): evaluated by the step kernel's OSC phase in a zero-step launch."""
        tau, _ = self.physics.osc_compute()
        tau = tau.astype(np.float64)
        return tau[0] if self.physics.num_envs == 1 else tau

    def current_position_error(self):
        _, eef, _ = self.physics.sites()
        return self.eef_target_position - eef[:, :3]

    def current_orientation_error(self):
        _, eef, _ = self.physics.sites()
        q, t = eef[:, 3:7].astype(np.float64), self.eef_target_quat
        qc = q * np.array([1, -1, -1, -1.0])
        w = t[:, 0] * qc[:, 0] - (t[:, 1:] * qc[:, 1:]).sum(1)
        v = (t[:, :1] * qc[:, 1:] + qc[:, :1] * t[:, 1:] + np.cross(t[:, 1:], qc[:, 1:]))
        return np.sign(w)[:, None] * v

    def is_converged(self):
        ep = np.linalg.norm(self.current_position_error(), axis=1)
        eo = np.linalg.norm(self.current_orientation_error(), axis=1)
        conv = (ep < self.position_threshold) & (eo < self.orientation_threshold)
        return bool(conv[0]) if len(conv) == 1 else conv
