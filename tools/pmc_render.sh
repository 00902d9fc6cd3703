#!/bin/bash
# k_render HBM counters at 256 envs (one pass per counter; a 4096-env dispatch with counters on was killed by
# the box's silence guard in round 1): tools/pmc_render.sh TAG
set -e
T=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/bench_render.py --envs 256 --frames 10 > gpurun_out/render_${T}_256.json 2>/dev/null
python tools/bench_render.py --envs 4096 --frames 10 > gpurun_out/render_${T}_4096.json 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_render_${T}_$C -- python tools/bench_render.py --envs 256 --frames 4 > gpurun_out/pmc_render_${T}_$C.log 2>&1
done
python - <<PY
import csv, glob, json
out = {"envs": 256, "resolution": [480, 640], "pixels": 256 * 480 * 640, "command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python tools/bench_render.py --envs 256 --frames 4 (one pass per counter)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_render_${T}_{c}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_render" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c + "_KiB_per_launch"] = sum(v) / len(v)
    out[c + "_launches"] = len(v)
    out[c.split("_")[0].lower() + "_bytes_per_pixel"] = sum(v) / len(v) * 1024.0 / out["pixels"]
out["bench_256"] = json.load(open("gpurun_out/render_${T}_256.json"))
out["bench_4096"] = json.load(open("gpurun_out/render_${T}_4096.json"))
json.dump(out, open("gpurun_out/render_${T}_hbm_counters.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("bench")}))
print(out["bench_256"]["ms_per_batch_frame"], out["bench_4096"]["ms_per_batch_frame"], out["bench_4096"]["roofline"]["frac"])
PY
