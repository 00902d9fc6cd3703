"""Copy one round's rocprofv3 outputs from gpurun_out/ (scratch) into profiles/ (tracked).

usage: python tools/summarize_profiles.py TAG STATS_DIR FETCH_DIR WRITE_DIR SQ_DIR [BUILD NOTE]
  STATS_DIR : rocprofv3 --kernel-trace --stats --output-format csv -d STATS_DIR -- python bench.py ...
  FETCH_DIR / WRITE_DIR / SQ_DIR : the three separate --pmc passes (never combined with traces
  other than --kernel-trace).
Writes profiles/TAG_kernel_stats.csv, TAG_pmc_{fetch,write,sq}.csv (k_step rows only) and
TAG_pmc_summary.json (per-launch means; FETCH_SIZE / WRITE_SIZE are KiB on gfx950).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "mre::k_step"


def one(pattern):
    f = sorted(glob.glob(pattern, recursive=True))
    if not f:
        raise SystemExit(f"nothing matches {pattern}")
    return f[0]


def counter_rows(d):
    path = one(os.path.join(d, "**", "*counter_collection.csv"))
    with open(path) as f:
        rows = [r for r in csv.DictReader(f) if r["Kernel_Name"].startswith(KERNEL)]
    return path, rows


def main():
    tag, stats_d, fetch_d, write_d, sq_d = sys.argv[1:6]
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    out = os.path.join(ROOT, "profiles")
    shutil.copy(one(os.path.join(stats_d, "**", "*kernel_stats.csv")), os.path.join(out, f"{tag}_kernel_stats.csv"))
    summary = {
        "command": "rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python bench.py --steps 10 "
                   "--warmup 2 --no-cpu-baseline (three separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*)",
        "kernel": KERNEL, "launch": "1 control tick = 5 physics steps x 4096 envs", "build": note}
    for name, d in (("fetch", fetch_d), ("write", write_d), ("sq", sq_d)):
        path, rows = counter_rows(d)
        with open(os.path.join(out, f"{tag}_pmc_{name}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        acc = defaultdict(list)
        for r in rows:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            summary[f"{k}_per_launch"] = sum(v) / len(v)
        summary["LDS_Block_Size"] = int(rows[0]["LDS_Block_Size"])
        summary["VGPR_Count"] = int(rows[0]["VGPR_Count"])
        summary["Accum_VGPR_Count"] = int(rows[0]["Accum_VGPR_Count"])
        summary["Scratch_Size"] = int(rows[0]["Scratch_Size"])
    summary["hbm_read_bytes_per_launch"] = summary["FETCH_SIZE_per_launch"] * 1024.0
    summary["hbm_write_bytes_per_launch"] = summary["WRITE_SIZE_per_launch"] * 1024.0
    summary["traffic_bytes_per_launch"] = summary["hbm_read_bytes_per_launch"] + summary["hbm_write_bytes_per_launch"]
    summary["note"] = ("FETCH_SIZE/WRITE_SIZE are in KiB. The gfx950 x2 FETCH_SIZE correction applies to 16-B-per-lane "
                       "streaming reads; this kernel reads dword rows, so the read figure is reported uncorrected "
                       "(upper bound with the correction: 2x).")
    with open(os.path.join(out, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
