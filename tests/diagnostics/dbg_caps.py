"""Where do the contact capacities overflow during a scripted pick+place? (chunked run_controller)"""
import sys, collections, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, colour_separator_task_config, PICK_HEIGHT, PRE_PICK_HEIGHT, home_quat
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=N)
env.reset()
prog, pick, place = env.sort_colours()
c = env._robot.arm_controller
mx = np.zeros((N, 2), int)
first = np.full(N, "", dtype=object)
def run(name, dur, chunk=0.1):
    global mx
    for k in range(int(round(dur / chunk))):
        env._robot.run_controller(chunk)
        st = env.physics.solver_stats()
        mx = np.maximum(mx, st[:, :2])
        ov = (env.physics.status() & 4) != 0
        new = ov & (first == "")
        first[new] = name
    st = env.physics.solver_stats()
    print(f"{name:10s} ncon mean {st[:,0].mean():5.1f} max {st[:,0].max():3d} | nefc mean {st[:,1].mean():5.1f} max {st[:,1].max():3d} | overflow so far {(first != '').sum()}")
for tag, pose in (("pick", pick), ("place", place)):
    p = pose.copy(); p[:, 2] = PICK_HEIGHT; pre = p.copy(); pre[:, 2] = PRE_PICK_HEIGHT
    c.set_target(position=pre[:, :3], velocity=np.zeros(3), quat=pre[:, 3:], angular_velocity=np.zeros(3)); run(tag + ":pre", 2.0)
    c.set_target(position=p[:, :3]); run(tag + ":down", 2.0)
    env._robot.end_effector_controller.status = "max" if tag == "pick" else "min"; run(tag + ":grip", 1.0)
    c.set_target(position=pre[:, :3]); run(tag + ":up", 2.0)
    c.set_target(position=env.eef_home_pose, quat=home_quat()); run(tag + ":home", 2.0)
print("first overflow phase:", dict(collections.Counter(first.tolist())))
print("per-env max (sampled every 0.1 s): ncon hist", np.bincount(np.minimum(mx[:, 0], 40))[8:], "nefc max", mx[:, 1].max())
