"""Trace the controller-site / pinch-site height during the scripted pick phases (chunked run_controller)."""
import sys, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, colour_separator_task_config, PICK_HEIGHT, PRE_PICK_HEIGHT
N = 4
env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=N)
env.reset()
prog, pick, place = env.sort_colours()
p = pick.copy(); p[:, 2] = PICK_HEIGHT; pre = p.copy(); pre[:, 2] = PRE_PICK_HEIGHT
c = env._robot.arm_controller
np.set_printoptions(precision=4, suppress=True, linewidth=220)
def run(name, dur, chunk=0.05):
    n = int(round(dur / chunk))
    rows = []
    for k in range(n):
        env._robot.run_controller(chunk)
        tcp, eef, props = env.physics.sites()
        q = env.physics.state()[0] if hasattr(env.physics, "state") else None
        rows.append((round((k + 1) * chunk, 3), eef[0].copy(), tcp[0].copy()))
    print(name)
    for t, e, tc in rows[:: max(1, n // 20)]:
        print("  t %.2f eef %s tcp %s" % (t, e, tc))
    print("  final eef", rows[-1][1], "min eef z", min(r[1][2] for r in rows), "min tcp z", min(r[2][2] for r in rows))
print("target pre", pre[0], "pick", p[0])
c.set_target(position=pre[:, :3], velocity=np.zeros(3), quat=pre[:, 3:], angular_velocity=np.zeros(3))
run("pre-pick", 2.0)
c.set_target(position=p[:, :3])
run("descend", 2.0)
env._robot.end_effector_controller.status = "max"
run("close", 1.0)
c.set_target(position=pre[:, :3])
run("lift", 2.0)
tcp, eef, props = env.physics.sites()
print("cube poses env0", props[0, :env.nprops[0]], "status", env.physics.status())
