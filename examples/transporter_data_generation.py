#!/usr/bin/env python3
"""BASELINE.json configs[4]: the reference's data-generation loop
(mujoco_robot_environments/transporter_network_data_generation.py:97-143) driving thousands of
RearrangementEnv instances on one MI355X.  envlogger/TFDS writing is stubbed (out of scope,
SURVEY.md section 8f): the "dataset" is the list of (pick, place) actions with their pixel coordinates; observations are zero images of the
reference shapes, or the batched overhead camera's images with --render.

    python examples/transporter_data_generation.py --num-envs 8192 --max-steps 2
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
    for step in range(max_steps):
        in_progress, pick_pose, place_pose = env.sort_colours()
        if not in_progress.any():
            print("Task demonstration is complete")
            break
        pick_action = {"pose": pick_pose, "pixel_coords": env.world_2_pixel(cam, pick_pose[:, :3]), "gripper_rot": 0.0}
        place_action = {"pose": place_pose, "pixel_coords": env.world_2_pixel(cam, place_pose[:, :3]), "gripper_rot": 0.0}
        ts = env.step(pick_action)
        if logger:
            logger.step(pick_action, ts, in_progress)
        ts = env.step(place_action)
        if logger:
            logger.step(place_action, ts, in_progress)
        obs = ts.observation
        episodes.append((pick_action, place_action))
        if args.render and step == 0:
            d = obs["overhead_camera/depth"]
            print(f"observation: rgb {tuple(obs['overhead_camera/rgb'].shape)} depth {tuple(d.shape)} "
                  f"(min {float(d.min()):.3f} m, max {float(d.max()):.3f} m) on {d.device}")
        nsim = 2 * 9000
        print(f"pair {step}: {in_progress.sum()} envs in progress, all phases converged in "
              f"{int(env.last_converged.sum())}/{args.num_envs} envs, "
              f"{args.num_envs * (step + 1) * nsim / (time.time() - t0):.3g} env-steps/s so far (reset included)")
    if logger:
        logger.flush()
        print("episode shards:", writer.close())
    done = ~env.sort_colours()[0]
    print(f"{int(done.sum())}/{args.num_envs} envs have every cube in its colour's target after {len(episodes)} pairs; "
          f"intrinsics fx={metadata['intrinsics']['fx']:.1f}; wall {time.time() - t0:.1f} s")
    env.close()


if __name__ == "__main__":
    main()
