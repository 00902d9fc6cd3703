"""Per-phase time of the slowest envs of a late bench tick (diagnostic stamp builds 0, 1, 2; the run is
deterministic, so the three builds see the same envs).  usage: python tests/diagnostics/dbg_straggler_phases.py [tick]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DIAG = os.path.join(ROOT, "tools", "_diag")
NAMES = ["position+crb+factor", "velocity", "collide", "assemble", "control", "smooth", "solve", "integrate+io",
         "newton: setup", "newton: direction", "newton: direction (re-used)", "newton: search",
         "direction: init (M rows)", "direction: J'DJ (matrix cores)", "direction: tiles to rows", "direction: elimination + back solve",
         "position_stage: kinematics", "position_stage: comPos", "direction: cone rows (count)", "direction: quad rows (count)",
         "search: M v, J v", "search: line search evaluations", "search: move, update, J'f, gradient", "search: evaluations (count)"]
if "MRE_LIB" not in os.environ:
    tick = sys.argv[1] if len(sys.argv) > 1 else "200"
    vals = []
    for k in (0, 1, 2, 4, 5, 7):
        env = dict(os.environ, MRE_LIB=os.path.join(DIAG, f"libmre_stamps{k}.so"))
        out = subprocess.check_output([sys.executable, __file__, tick], env=env)
        vals.append(np.frombuffer(out[-4096 * 4 * 4:], np.int32).reshape(4096, 4))
    v = np.concatenate(vals, axis=1).astype(np.float64)   # [N, 20]
    tot = v[:, :12].sum(axis=1)
    order = np.argsort(-tot)
    top = order[:40]
    print(f"tick {tick}: total per env mean {tot.mean():.0f} max {tot.max():.0f}; slowest 40 envs mean {tot[top].mean():.0f}")
    for j, n in enumerate(NAMES):
        print(f"{n:32s} all envs {v[:, j].mean():8.0f}   slowest 40 {v[top, j].mean():8.0f}   ratio {v[top, j].mean() / max(v[:, j].mean(), 1):5.2f}")
else:
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from mujoco_robot_environments_amd import rng, lib as _lib
    from mujoco_robot_environments_amd.physics import BatchedPhysics
    N, tick = 4096, int(sys.argv[1])
    phys = BatchedPhysics(N, solver="Newton"); ids = np.arange(N)
    bench.setup_envs(phys, 0, ids)
    seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(tick + 1)).astype(np.float32)).to(phys.device).contiguous()
    for t in range(tick + 1):
        phys.rollout(seq[t:t + 1], control_steps=5)
    st = np.empty((N, 4), np.int32)
    _lib.check(_lib.lib().mre_get_solver_stats(phys._h, st.ctypes.data), "stats")
    sys.stdout.flush()
    sys.stdout.buffer.write(st.tobytes())
