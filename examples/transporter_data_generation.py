#!/usr/bin/env python3
"""BASELINE.json configs[4]: the reference's data-generation loop
(mujoco_robot_environments/transporter_network_data_generation.py:97-143) driving thousands of
RearrangementEnv instances on one MI355X for ``dataset.max_steps`` pick / place pairs (config/dataset/default.yaml).
Observations are zero images of the reference shapes, or the batched overhead camera's images with --render; with
--out the episodes of the first --log-envs envs are written as RLDS / TFDS shards (mujoco_robot_environments_amd/
dataset.py) in place of envlogger's TFDSBackendWriter.

    python examples/transporter_data_generation.py --num-envs 8192 --render --out /tmp/shards
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mujoco_robot_environments_amd.tasks.rearrangement import (  # noqa: E402
    BatchedRearrangementEnv, colour_separator_task_config)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=8192)
    ap.add_argument("--max-steps", type=int, default=None, help="pick/place pairs per episode (config: dataset.max_steps)")
    ap.add_argument("--gains", type=float, nargs=6, default=None, metavar=("KP_POS", "KD_POS", "KP_ORI", "KD_ORI", "KP_NULL", "KD_NULL"),
                    help="OSC gains instead of osc.yaml's (config.TUNED_OSC_GAINS: 525.4 215.2 766.6 158.6 22.6 7.2)")
    ap.add_argument("--tuned-gains", action="store_true", help="config.apply_tuned_osc_gains (CMA-ES result for this arm model)")
    ap.add_argument("--render", action="store_true", help="overhead-camera observations (depth + RGB, CUDA tensors) instead of zero images")
    ap.add_argument("--out", default=None, help="write the episodes as RLDS shards (TFRecord, the reference ds_config's feature "
                    "keys: mujoco_robot_environments_amd/dataset.py) into this directory")
    ap.add_argument("--log-envs", type=int, default=64, help="with --out: episodes of the first K envs are written")
    ap.add_argument("--solver", choices=["Newton", "PGS"], default="Newton", help="constraint solver (the reference's MuJoCo runs Newton)")
    args = ap.parse_args()
    cfg = colour_separator_task_config()
    if args.tuned_gains:
        from mujoco_robot_environments_amd.config import apply_tuned_osc_gains
        apply_tuned_osc_gains(cfg)
    if args.gains:
        g = cfg.robots.arm.controller_config.controller_params.gains
        g.position.kp, g.position.kd, g.orientation.kp, g.orientation.kd, g.nullspace.kp, g.nullspace.kd = args.gains
    max_steps = args.max_steps or cfg.dataset.max_steps
    env = BatchedRearrangementEnv(cfg=cfg, num_envs=args.num_envs, render=args.render, solver=args.solver)
    cam = "overhead_camera/overhead_camera"
    t0 = time.time()
    ts = env.reset()
    obs = ts.observation
    metadata = env.get_camera_metadata()  # episode metadata of the reference (calibration_metadata)
    episodes = []
    logger = None
    if args.out:
        from mujoco_robot_environments_amd.dataset import BatchedEpisodeLogger, EpisodeWriter
        writer = EpisodeWriter(args.out, f"{cfg.get('name', 'colour_splitter')}", env.overhead_camera_height, env.overhead_camera_width,
                               max_episodes_per_file=cfg.dataset.max_episodes_per_file)
        mask = np.arange(args.num_envs) < args.log_envs
        logger = BatchedEpisodeLogger(env, writer, mask)
        logger.reset(ts)
    print(f"reset: {time.time() - t0:.1f} s; placement failed in {int(env.placement_failed.sum())} envs, "
          f"not settled in {int(env.not_settled.sum())}")
    # `alive`: the episode is still running.  The reference's step() raises when a scripted phase does not converge
    # (tasks/rearrangement.py:371-440) and the loop drops out of the episode ("Task demonstration failed with
    # exception", transporter_network_data_generation.py:137-139); here such an env stops acting and logging.
    alive = ~env.placement_failed
    steps_logged = np.zeros(args.num_envs, np.int64)
    t_phys = t_log = 0.0
    for step in range(max_steps):
        in_progress, pick_pose, place_pose = env.sort_colours()
        active = in_progress & alive
        if not active.any():
            print("Task demonstration is complete")
            break
        pick_action = {"pose": pick_pose, "pixel_coords": env.world_2_pixel(cam, pick_pose[:, :3]), "gripper_rot": 0.0}
        place_action = {"pose": place_pose, "pixel_coords": env.world_2_pixel(cam, place_pose[:, :3]), "gripper_rot": 0.0}
        env.last_converged[:] = True
        env.failed_phase[:] = ""
        for action in (pick_action, place_action):
            t1 = time.time()
            ts = env.step(action)
            env.physics.sync()
            t2 = time.time()
            if logger:
                logger.step(action, ts, active)
            steps_logged += active
            t_phys += t2 - t1
            t_log += time.time() - t2
        obs = ts.observation
        episodes.append((pick_action, place_action))
        if args.render and step == 0:
            d = obs["overhead_camera/depth"]
            print(f"observation: rgb {tuple(obs['overhead_camera/rgb'].shape)} depth {tuple(d.shape)} "
                  f"(min {float(d.min()):.3f} m, max {float(d.max()):.3f} m) on {d.device}")
        failed_now = active & ~env.last_converged
        names, counts = np.unique(env.failed_phase[failed_now].astype(str), return_counts=True)
        names = [str(n) for n in names]
        alive &= env.last_converged | ~active   # (an idle env's no-op phases do not end an episode)
        nsim = 2 * 9000
        still = env.sort_colours(peek=True)[0]
        print(f"pair {step}: {int(in_progress.sum())} envs in progress, {int(active.sum())} of them acted; "
              f"a phase did not converge in {int(failed_now.sum())} "
              f"({', '.join(f'{n!r}: {c}' for n, c in zip(names, counts)) or 'none'}); "
              f"sorted so far {int((~still).sum())}/{args.num_envs}; "
              f"{args.num_envs * (step + 1) * nsim / (time.time() - t0):.3g} env-steps/s so far "
              f"(reset, camera{', shard logging' if logger else ''} included)", flush=True)
    if logger:
        t1 = time.time()
        logger.flush()
        print("episode shards:", writer.close())
        t_log += time.time() - t1
    done = ~env.sort_colours(peek=True)[0]
    n_pairs = len(episodes)
    print(f"{int(done.sum())}/{args.num_envs} envs have every cube in its colour's target after {n_pairs} pairs "
          f"({int((done & alive).sum())} of them with every phase converged -- the episodes the reference's loop would keep); "
          f"steps per env episode: min {int(steps_logged.min())} median {int(np.median(steps_logged))} max {int(steps_logged.max())}; "
          f"intrinsics fx={metadata['intrinsics']['fx']:.1f}; wall {time.time() - t0:.1f} s "
          f"(step() incl. camera {t_phys:.1f} s, logging + shards {t_log:.1f} s); "
          f"{args.num_envs * n_pairs * 18000 / max(t_phys, 1e-9):.3g} env-steps/s inside step()")
    env.close()


if __name__ == "__main__":
    main()
