"""Per-phase share of a control tick, from diagnostic builds with s_memtime stamps.

  python tools/phase_stamps.py build      # here (no GPU): tools/_diag/libmre_stamps{0,1}.so
  python tools/phase_stamps.py run [N]    # on the GPU box; MRE_LIB selects the diagnostic library
The product library never carries the stamps (MRE_PHASE_STAMPS is undefined in lib.build()).
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mujoco_robot_environments_amd", "csrc")
DIAG = os.path.join(ROOT, "tools", "_diag")
NAMES = ["position+crb+factor", "velocity", "collide", "assemble", "control", "smooth", "solve", "integrate+io"]

if sys.argv[1] == "build":
    os.makedirs(DIAG, exist_ok=True)
    for k in (0, 1):
        base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
        objs = []
        for name, src, flags in (("k", "mre_kernels.hip", [f"-DMRE_PHASE_STAMPS={k}"]),
                                 ("kl", "mre_kernels.hip", ["-DMRE_LARGE_CAPS", f"-DMRE_PHASE_STAMPS={k}"]), ("r", "mre_render.hip", []),
                                 ("api", "mre_api.cpp", [])):
            objs.append(os.path.join(DIAG, f"{name}{k}.o"))
            subprocess.check_call(base + flags + ["-c", os.path.join(CSRC, src), "-o", objs[-1]])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(DIAG, f"libmre_stamps{k}.so")] + objs)
elif sys.argv[1] == "run":
    if "MRE_LIB" not in os.environ:
        tot = []
        for k in (0, 1):
            env = dict(os.environ, MRE_LIB=os.path.join(DIAG, f"libmre_stamps{k}.so"))
            out = subprocess.check_output([sys.executable, __file__, "run"] + sys.argv[2:], env=env, text=True)
            tot += [float(x) for x in out.split()[-4:]]
        s = sum(tot)
        for n, v in zip(NAMES, tot):
            print(f"{n:22s} {v:12.0f}  {100 * v / s:5.1f} %")
    else:
        sys.path.insert(0, ROOT)
        import numpy as np, torch
        import bench
        from mujoco_robot_environments_amd import rng
        from mujoco_robot_environments_amd.physics import BatchedPhysics
        N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
        phys = BatchedPhysics(N); ids = np.arange(N)
        bench.setup_envs(phys, 0, ids)
        seq = torch.from_numpy(rng.random_actions(0, ids, np.arange(20)).astype(np.float32)).to(phys.device).contiguous()
        acc = np.zeros(4)
        for t in range(20):
            phys.rollout(seq[t:t + 1], control_steps=5)
            acc += phys.solver_stats().astype(np.float64).mean(axis=0)
        print(*(acc / 20))
