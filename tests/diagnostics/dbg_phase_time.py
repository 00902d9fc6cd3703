"""Wall time of each scripted pick / place phase (tuned gains), with capacity-fallback counters."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.tasks.rearrangement import (BatchedRearrangementEnv, colour_separator_task_config,
                                                                PICK_HEIGHT, PRE_PICK_HEIGHT, home_quat)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = colour_separator_task_config()
g = cfg.robots.arm.controller_config.controller_params.gains
g.position.kp, g.position.kd, g.orientation.kp, g.orientation.kd, g.nullspace.kp, g.nullspace.kd = 221.2, 59.5, 543.0, 169.6, 81.6, 36.2
env = BatchedRearrangementEnv(cfg=cfg, num_envs=N)
t = time.time(); env.reset(); print(f"reset {time.time()-t:.2f} s")
t = time.time(); prog, pick, place = env.sort_colours(); print(f"sort_colours {time.time()-t:.2f} s")
c = env._robot.arm_controller
def run(name, dur):
    f0 = env.physics.fallback_stats()
    t = time.time(); conv = env._robot.run_controller(dur); dt = time.time() - t
    f1 = env.physics.fallback_stats()
    st = env.physics.solver_stats()
    print(f"{name:12s} {dt:6.2f} s  {N*dur*1000/dt/1e6:5.2f} M env-steps/s | converged {conv.mean():.2f} | ncon {st[:,0].mean():.1f} nefc {st[:,1].mean():.1f} | "
          f"large {f1['large_envs']} reruns +{f1['reruns']-f0['reruns']}")
for tag, pose in (("pick", pick), ("place", place)):
    p = pose.copy(); p[:, 2] = PICK_HEIGHT; pre = p.copy(); pre[:, 2] = PRE_PICK_HEIGHT
    c.set_target(position=pre[:, :3], velocity=np.zeros(3), quat=pre[:, 3:], angular_velocity=np.zeros(3)); run(tag + ":pre", 2.0)
    c.set_target(position=p[:, :3]); run(tag + ":down", 2.0)
    env._robot.end_effector_controller.status = "max" if tag == "pick" else "min"; run(tag + ":grip", 1.0)
    c.set_target(position=pre[:, :3]); run(tag + ":up", 2.0)
    c.set_target(position=env.eef_home_pose, quat=home_quat()); run(tag + ":home", 2.0)
