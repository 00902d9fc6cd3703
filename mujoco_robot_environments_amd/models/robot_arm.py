"""RobotArm: mirror of the reference's ``models/robot_arm.py`` over a batch of envs.

``run_controller(duration)`` is THE hot loop (reference models/robot_arm.py:61-94): per
5 ms control tick one OSC torque + one gripper command, then ``control_steps`` physics
steps.  Here the whole call is ONE kernel launch (mre_run_controller); the number of ticks
replicates the reference's ``while physics.data.time - start_time < duration`` test on an
fp64 clock advanced by the timestep once per physics step, and -- as in the reference,
where ``gripper_converged`` can never become True with a gripper attached (:65,86-88) --
the loop always runs the full duration."""
from __future__ import annotations

import numpy as np

from ..controllers import OSC, MinMax


class RobotArm:
    def __init__(self, physics, controller_params=None, gripper_cfg=None, gripper: bool = True, strict: bool = True):
        self.physics = physics
        self.arm_controller = OSC(physics, controller_config=controller_params)
        self.end_effector = "robotiq_2f85" if gripper else None
        mm = dict(min_val=0.0, max_val=255.0)
        if gripper_cfg is not None:
            mm = dict(min_val=float(gripper_cfg.get("min_val", 0.0)), max_val=float(gripper_cfg.get("max_val", 255.0)))
        self.end_effector_controller = MinMax(num_envs=physics.num_envs, physics=physics, **mm)
        control_dt = 0.005 if controller_params is None else float(controller_params["control_dt"])
        self.timestep = physics.timestep
        # robot_arm.py:53
        self.control_steps = int(control_dt // self.timestep)
        if self.control_steps < 1 and strict:
            raise ValueError("control_dt < physics_dt gives control_steps = 0: the reference's "
                             "run_controller would never advance time (SURVEY.md App. D.9)")
        self.time = 0.0  # physics.data.time (fp64, += timestep per step)

    @property
    def eef_pose(self):
        """site_xpos of the gripper pinch site (robot_arm.py:55-58)."""
        tcp, _, _ = self.physics.sites()
        return tcp[0].astype(np.float64) if self.physics.num_envs == 1 else tcp.astype(np.float64)

    def ticks_for(self, duration: float) -> int:
        t, start, ticks = self.time, self.time, 0
        while t - start < duration:
            for _ in range(self.control_steps):
                t += self.timestep
            ticks += 1
        return ticks

    def run_controller(self, duration: float):
        if self.control_steps < 1:  # LasaDrawEnv's config (physics_dt 0.01 > control_dt 0.005): the reference spins forever
            raise ValueError("control_steps = 0: run_controller would never advance time (SURVEY.md App. D.9)")
        ticks = self.ticks_for(duration)
        conv = self.physics.run_controller(ticks, self.control_steps)
        for _ in range(ticks * self.control_steps):
            self.time += self.timestep
        return bool(conv[0]) if self.physics.num_envs == 1 else conv
