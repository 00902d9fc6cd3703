"""Which envs are slow late in the random-action run?  Run twice (deterministic): once with the
stamps build (per-env solve cycles), once with the product build (per-env ncon/nefc)."""
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
T = int(os.environ.get("T", "200"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, bench
    from mujoco_robot_environments_amd import rng
    from mujoco_robot_environments_amd.physics import BatchedPhysics
    N = 4096
    phys = BatchedPhysics(N); ids = np.arange(N)
    phys.set_fallback(False)
    bench.setup_envs(phys, 0, ids)
    seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(T)).astype(np.float32)).to(phys.device).contiguous()
    for t in range(T):
        phys.rollout(seq[t:t + 1], control_steps=5)
    np.save(sys.argv[2], np.concatenate([phys.solver_stats(), phys.qpos()[:, :15]], axis=1))
else:
    out = {}
    for name, lib in (("stamps", os.path.join(ROOT, "tools", "_diag", "libmre_stamps1.so")), ("stats", None)):
        env = dict(os.environ)
        if lib: env["MRE_LIB"] = lib
        f = f"/tmp/tail_{name}.npy"
        subprocess.check_call([sys.executable, __file__, "child", f], env=env)
        out[name] = np.load(f)
    cyc = out["stamps"][:, :4] * 16.0   # control, smooth, solve, integrate buckets
    st = out["stats"]
    solve = cyc[:, 2]
    order = np.argsort(-solve)
    print(f"tick {T}: solve cycles per env: median {np.median(solve):.3g} p99 {np.percentile(solve, 99):.3g} max {solve.max():.3g}")
    np.set_printoptions(precision=2, suppress=True, linewidth=200)
    for i in order[:8]:
        print(f"  env {i}: solve {solve[i]:.3g} cycles ({solve[i]/np.median(solve):.1f}x median) ncon {int(st[i,0])} nefc {int(st[i,1])} nl {int(st[i,3])} arm q {st[i,4:11]} fingers {st[i,11:19]}")
