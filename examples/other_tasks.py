"""The reference's three small environments, batched: BaseEnv (tasks/base.py), PushEnv (tasks/push.py) and
LasaDrawEnv (tasks/lasa_draw.py) on the same step kernels as RearrangementEnv.

    python examples/other_tasks.py --num-envs 4096 --ticks 200

Each env runs its own hot loop (one command, five physics steps) for --ticks ticks:
  push  -- the OSC target sweeps the tool through the block (envs offset in y: centred pushes ... misses);
  lasa  -- deployment config (position actuators, physics_dt 0.01): joint targets on a circle in joint space;
  base  -- the mocap target is lowered to the floor, the gripper closes half way.
Prints env-steps/s (1 env-step = one physics step of one env) with the launches and the per-tick host work included.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mujoco_robot_environments_amd import config as cfgm  # noqa: E402
from mujoco_robot_environments_amd.tasks.base import BatchedBaseEnv  # noqa: E402
from mujoco_robot_environments_amd.tasks.lasa_draw import BatchedLasaDrawEnv  # noqa: E402
from mujoco_robot_environments_amd.tasks.push import BatchedPushEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--solver", choices=["Newton", "PGS"], default="Newton")
    args = ap.parse_args()
    N, T = args.num_envs, args.ticks
    ys = np.linspace(0.0, 0.07, N)

    env = BatchedPushEnv(num_envs=N, solver=args.solver)
    env.reset()
    env.physics.sync()
    t0 = time.perf_counter()
    for k in range(T):
        mp = np.stack([np.full(N, 0.2 + 0.0006 * k * 300 / T), ys, np.full(N, 0.53 - 0.175)], axis=1)
        env.interactive_tuning(mocap_pos=mp)
    x = env.block_pose()[:, 0]
    dt = time.perf_counter() - t0
    li = env.physics.launch_info()
    print(f"push  {N} envs x {T} ticks: {N * T * 5 / dt / 1e6:.2f} M env-steps/s; block moved > 1 cm in "
          f"{int((x > 0.31).sum())} envs, untouched in {int((np.abs(x - 0.3) < 1e-3).sum())}; last tick: max contacts "
          f"{int(li['ncon'].max())}, max rows {int(li['nefc'].max())}, {int(li['overflow'].sum())} envs on the large kernel")
    env.close()

    env = BatchedLasaDrawEnv(cfg=cfgm.lasa_deployment_config(), num_envs=N, solver=args.solver)
    env.reset()
    home = np.asarray(env.model["home_qpos"])
    amp = np.linspace(0.1, 0.3, N)[:, None]
    env.physics.sync()
    t0 = time.perf_counter()
    for k in range(T):
        tgt = home[None] + amp * np.sin(0.05 * k + np.arange(7))[None]
        q = env.move_to_joint_position_target(tgt)
    dt = time.perf_counter() - t0
    print(f"lasa  {N} envs x {T} ticks (position actuators, physics_dt 0.01): {N * T * 5 / dt / 1e6:.2f} M env-steps/s; "
          f"max |q - target| {np.abs(q - tgt).max():.4f} rad")
    env.close()

    env = BatchedBaseEnv(num_envs=N, solver=args.solver)
    env.reset()
    env.physics.sync()
    t0 = time.perf_counter()
    for k in range(T):
        if k == T // 2:
            env._robot.end_effector_controller.status = "max"
        mp = np.tile([0.4, 0.0, 0.2 - 0.23 * min(k / (0.8 * T), 1.0)], (N, 1))
        mp[:, 1] = ys
        env.interactive_tuning(mocap_pos=mp)
    qp = env.physics.qpos()
    dt = time.perf_counter() - t0
    tcp, _, _ = env.physics.sites()
    print(f"base  {N} envs x {T} ticks: {N * T * 5 / dt / 1e6:.2f} M env-steps/s; pinch site z {tcp[:, 2].min():.3f}.."
          f"{tcp[:, 2].max():.3f}, driver joints {qp[:, 7].min():.2f}..{qp[:, 7].max():.2f} rad")
    env.close()


if __name__ == "__main__":
    main()
