"""Follow held cubes through the scripted place phases."""
import sys, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, colour_separator_task_config, PICK_HEIGHT, PRE_PICK_HEIGHT, home_quat
N = 64
env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=N)
env.reset()
np.set_printoptions(precision=4, suppress=True, linewidth=200)
poses0 = env.physics.sites()[2].copy()
prog, pick, place = env.sort_colours()
which = np.array([int(np.argmin(np.linalg.norm(poses0[i, :env.nprops[i], :2] - pick[i, :2], axis=1))) for i in range(N)])
env.step({"pose": pick.copy()})
p1 = env.physics.sites()[2]
held = np.nonzero(p1[np.arange(N), which, 2] > 0.5)[0]
print("held envs", held[:10], "of", N)
sel = held[:3]
c = env._robot.arm_controller
p = place.copy(); p[:, 2] = PICK_HEIGHT; pre = p.copy(); pre[:, 2] = PRE_PICK_HEIGHT
def report(tag):
    tcp, eef, props = env.physics.sites()
    st = env.physics.state() if hasattr(env.physics, "state") else None
    for i in sel:
        print(f"  {tag:12s} env {i}: tcp {tcp[i]} cube {props[i, which[i], :3]} rel {props[i, which[i], :3] - tcp[i]} fingers {env.physics.qpos()[i, 7]:.3f}")
report("after pick")
def run(name, dur, chunk=0.25):
    for k in range(int(round(dur / chunk))):
        env._robot.run_controller(chunk)
        report(f"{name}@{(k+1)*chunk:.2f}")
c.set_target(position=pre[:, :3], quat=pre[:, 3:]); run("pre-place", 2.0, 0.5)
c.set_target(position=p[:, :3]); run("descend", 2.0, 0.5)
env._robot.end_effector_controller.status = "min"; run("open", 1.0, 0.25)
c.set_target(position=pre[:, :3]); run("up", 2.0, 1.0)
print("place targets", place[sel, :3])
