"""MinMax gripper controller: host mirror of ``mujoco_controllers.min_max.MinMax``
(third-party, absent from the reference checkout; API from its call sites:
models/robot_arm.py:73, tasks/rearrangement.py:380,422,
config/robots/end_effector/controller_config/min_max.yaml:1-4)."""
from __future__ import annotations

import numpy as np


class MinMax:
    def __init__(self, min_val: float = 0.0, max_val: float = 255.0, num_envs: int = 1, physics=None):
        self.min_val, self.max_val = float(min_val), float(max_val)
        self._closed = np.zeros(num_envs, np.uint8)
        self._physics = physics
        self._status = "min"

    @property
    def status(self):
        return self._status

    @status.setter
    def status(self, value):
        """'max' (close) / 'min' (open); a per-env sequence of those is accepted for batches."""
        if isinstance(value, str):
            assert value in ("min", "max")
            self._status = value
            self._closed[:] = 1 if value == "max" else 0
        else:
            vals = list(value)
            assert len(vals) == len(self._closed)
            self._closed[:] = [1 if v == "max" else 0 for v in vals]
            self._status = vals
        if self._physics is not None:
            self._physics.gripper_set(self._closed)

    def compute_control_output(self):
        out = np.where(self._closed > 0, self.max_val, self.min_val)
        return float(out[0]) if len(out) == 1 else out
