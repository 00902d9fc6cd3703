"""configs[2] diagnostic: one scripted pick+place per env; per-phase convergence and placement success."""
import sys, time, collections, numpy as np
sys.path.insert(0, '.')
from mujoco_robot_environments_amd.tasks.rearrangement import BatchedRearrangementEnv, colour_separator_task_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = colour_separator_task_config()
if len(sys.argv) > 3:   # optional OSC gain override: kp_pos kd_pos [kp_ori kd_ori kp_null kd_null]
    g = cfg.robots.arm.controller_config.controller_params.gains
    vals = [float(v) for v in sys.argv[3:]]
    g.position.kp, g.position.kd = vals[0], vals[1]
    if len(vals) >= 6:
        g.orientation.kp, g.orientation.kd, g.nullspace.kp, g.nullspace.kd = vals[2:6]
    print("OSC gains:", g)
env = BatchedRearrangementEnv(cfg=cfg, num_envs=N)
env.reset()
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for pair in range(npairs):
    poses0 = env.physics.sites()[2].copy()
    prog, pick, place = env.sort_colours()
    which = np.array([int(np.argmin(np.linalg.norm(poses0[i, :env.nprops[i], :2] - pick[i, :2], axis=1))) for i in range(N)])
    env.failed_phase[:] = ""
    t = time.time(); env.step({"pose": pick.copy()}); t_pick = time.time() - t
    p1 = env.physics.sites()[2].copy()
    conv_pick = env.last_converged.copy()
    fp_pick = collections.Counter(env.failed_phase.tolist())
    held = p1[np.arange(N), which, 2] > 0.5   # cube travels with the gripper to the home pose
    t = time.time(); env.step({"pose": place.copy()}); t_place = time.time() - t
    p2 = env.physics.sites()[2].copy()
    d = np.linalg.norm(p2[np.arange(N), which, :2] - place[:, :2], axis=1)
    moved_others = np.zeros(N)
    for i in range(N):
        o = [p for p in range(env.nprops[i]) if p != which[i]]
        if o: moved_others[i] = np.linalg.norm(p2[i, o, :2] - poses0[i, o, :2], axis=1).max()
    st = env.physics.status()
    print(f"pair {pair}: in progress {prog.sum()}/{N}; pick {t_pick:.1f}s place {t_place:.1f}s")
    print("  pick phases failing first:", dict(fp_pick))
    print("  place phases failing first:", dict(collections.Counter(env.failed_phase.tolist())))
    print(f"  cube held at home after pick: {(held & prog).sum()}/{prog.sum()}; placed within 3 cm: {((d < 0.03) & prog).sum()}/{prog.sum()}; "
          f"within 1 cm: {((d < 0.01) & prog).sum()}; other cubes disturbed > 1 cm: {(moved_others > 0.01).sum()}")
    print("  status: nan", ((st & 2) != 0).sum(), "overflow", ((st & 4) != 0).sum(), " final cube z range", p2[..., 2][p2[..., 2] > 0].min(), p2[..., 2].max())
    bad = np.nonzero(prog & ~held)[0][:5]
    for i in bad:
        print("   not held: env", i, "cube", which[i], "before", poses0[i, which[i], :3], "after pick", p1[i, which[i], :3], "pick pose", pick[i])
