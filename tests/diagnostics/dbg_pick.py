import sys, time, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, colour_separator_task_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
env = BatchedRearrangementEnv(cfg=colour_separator_task_config(), num_envs=N)
t = time.time(); ts = env.reset(); print("reset", time.time() - t, ts.step_type, {k: (v.shape, v.dtype) for k, v in ts.observation.items()})
print("nprops", env.nprops, "colours", env.prop_colours[:3])
poses0 = env.physics.sites()[2]
print("prop z after settle", poses0[:, :, 2].round(4)[:3])
prog, pick, place = env.sort_colours()
print("in progress", prog, "\npick", pick[:2].round(3), "\nplace", place[:2].round(3))
np.set_printoptions(precision=4, suppress=True, linewidth=200)
which = np.zeros(N, int)
for i in range(N):
    info = env.props_info_env(i, poses0)
    for pid, a in info.items():
        if np.allclose(a["position"][:2], pick[i, :2], atol=1e-6):
            which[i] = pid - 12
t = time.time()
env.step({"pose": pick.copy()})
print("pick time", time.time() - t, "converged", env.last_converged, "failed", list(env.failed_phase))
p1 = env.physics.sites()[2]
tcp, eef, _ = env.physics.sites()
for i in range(min(N, 4)):
    print(i, "picked prop", which[i], "before", poses0[i, which[i], :3], "after pick", p1[i, which[i], :3], "tcp", tcp[i])
t = time.time()
env.step({"pose": place.copy()})
print("place time", time.time() - t, "converged", env.last_converged, "failed", list(env.failed_phase))
p2 = env.physics.sites()[2]
ok = 0
for i in range(N):
    d = np.linalg.norm(p2[i, which[i], :2] - place[i, :2])
    ok += d < 0.03
    if i < 4:
        print(i, "after place", p2[i, which[i], :3], "target", place[i, :3], "xy err", d)
print("placed within 3cm:", ok, "/", N, "status", env.physics.status())
