#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the GPU box (tools/pmc_calibrate.hip): bash tools/pmc_calibrate.sh > gpurun_out/pmc_calibration.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_cal_fetch -- tools/_diag/pmc_calibrate > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_cal_write -- tools/_diag/pmc_calibrate > /dev/null 2>&1
python - <<'PY'
import csv, glob, json, collections
GIB = 1 << 30
ROWS = (GIB // 4 // 44) * 44 * 4
out = {"note": "bytes moved / (counter x 1024) per access pattern on gfx950: the factor a FETCH_SIZE / WRITE_SIZE reading of that "
               "pattern has to be multiplied by (MI355X_MICROARCH.md: 16-B-per-lane reads read 1/2, other widths uncalibrated)"}
for name, d in (("FETCH_SIZE", "gpurun_out/pmc_cal_fetch"), ("WRITE_SIZE", "gpurun_out/pmc_cal_write")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        kib = sum(v[1:]) / max(len(v) - 1, 1)      # (first repetition: cold)
        known = ROWS if "rows44" in k else GIB
        moves = ("read" in k and name == "FETCH_SIZE") or ("write" in k and name == "WRITE_SIZE") or "rows44" in k
        out[f"{name} {k}"] = {"counter_KiB": kib, "bytes_known": known if moves else 0,
                              "factor": (known / (kib * 1024.0)) if moves and kib > 0 else None}
print(json.dumps(out, indent=1))
PY
