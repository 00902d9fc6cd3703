#!/bin/bash
# Full measurement set of one build on the GPU box (run through gpurun; outputs under gpurun_out/):
#   tools/measure_round.sh TAG
# bench (default = PGS headline + Newton beside it, and the first 40 ticks), then per solver: kernel
# trace + stats and four separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_INSTS_VALU_*_F32).
# Counter passes carry --kernel-trace only (never sys / runtime / hip / hsa traces) and run the bench's default
# 20 warm-up + 200 timed launches, so that per-launch counter means and the timed launch duration describe the
# same launches (summarize_profiles.py drops the warm-up rows).
set -e
T=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_${T}_40.json 2>/dev/null
rocprofv3 -L > gpurun_out/counters_$T.txt 2>&1 || true
for S in PGS Newton; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_$S -- python bench.py --solver $S --no-cpu-baseline > gpurun_out/bench_${T}_prof_$S.json 2>/dev/null
  head -4 $(find gpurun_out/prof_${T}_$S -name "*kernel_stats.csv" | head -1)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_${T}_$S -- python bench.py --solver $S --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_${T}_$S -- python bench.py --solver $S --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq_${T}_$S -- python bench.py --solver $S --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_flop_${T}_$S -- python bench.py --solver $S --no-cpu-baseline > gpurun_out/pmc_flop_${T}_$S.log 2>&1 || echo "flop counter pass failed for $S (see gpurun_out/pmc_flop_${T}_$S.log)"
done
python -c "import json; d=json.load(open('gpurun_out/bench_$T.json')); print(d['value'], d['ms_per_step'], d['health'], d['newton']['value'], d['cpu_baseline']['value'])"
python -c "import json; d=json.load(open('gpurun_out/bench_${T}_40.json')); print(d['value'], d['ms_per_step'], d['newton']['value'])"
