"""A/B builds of libmre.so with extra compiler flags (diagnostics; the product build is lib.build()).

  python tools/build_variant.py NAME [-DMACRO ...]     # here (no GPU): tools/_diag/libmre_NAME.so
  MRE_LIB=tools/_diag/libmre_NAME.so python bench.py    # on the GPU box
Known switches: -DMRE_PGS_F32 (PGS: robot-contact block update in float32, as up to round 3)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get("MRE_CSRC", os.path.join(ROOT, "mujoco_robot_environments_amd", "csrc"))   # (MRE_CSRC: another checkout's sources, for A/B against an earlier commit)
DIAG = os.path.join(ROOT, "tools", "_diag")
name, extra = sys.argv[1], sys.argv[2:]
os.makedirs(DIAG, exist_ok=True)
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=on",
        "-fno-hip-fp32-correctly-rounded-divide-sqrt"] + extra
units = [("k", "mre_kernels.hip", []), ("kl", "mre_kernels.hip", ["-DMRE_LARGE_CAPS"]), ("kn", "mre_kernels.hip", ["-DMRE_NEWTON"]),
         ("kln", "mre_kernels.hip", ["-DMRE_LARGE_CAPS", "-DMRE_NEWTON"]), ("r", "mre_render.hip", []), ("api", "mre_api.cpp", [])]
procs = []
for u, src, flags in units:
    o = os.path.join(DIAG, f"{u}_{name}.o")
    procs.append((o, subprocess.Popen(base + flags + ["-c", os.path.join(CSRC, src), "-o", o])))
for o, p in procs:
    assert p.wait() == 0, o
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(DIAG, f"libmre_{name}.so")] + [o for o, _ in procs])
print(os.path.join(DIAG, f"libmre_{name}.so"))
