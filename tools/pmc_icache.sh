#!/bin/bash
# Instruction-cache counter pass (diagnostic): tools/pmc_icache.sh TAG
set -e
T=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for S in PGS Newton; do
  rocprofv3 --pmc ${PMC_SET:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES} --kernel-trace --output-format csv -d gpurun_out/pmc_ic_${T}_$S -- python bench.py --solver $S --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_ic_${T}_$S.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_ic_${T}_$S/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("mre::k_step(") or r["Kernel_Name"].startswith("mre::k_step_newton("):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$S", {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
