#!/bin/bash
# Full measurement set of one build on the GPU box (run through gpurun; outputs under gpurun_out/):
#   tools/measure_round.sh TAG
# bench (default = PGS headline + Newton beside it), then per solver and per bench argument set -- bench.py's own
# default (--steps 200 --warmup 20) AND the driver's end-of-round arguments (--steps 20 --warmup 5) -- a kernel
# trace + stats and five separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_INSTS_VALU_*_F32/F64 | MFMA ops).
# Counter passes carry --kernel-trace only (never sys / runtime / hip / hsa traces) and run the SAME bench arguments
# as the line they annotate, so that per-launch counter means and the timed launch duration describe the same
# launches (summarize_profiles.py drops the warm-up rows and records steps / warmup / source hash; bench.py quotes a
# summary only when all three match its own run).
set -e
T=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${T}_s20w5.json 2>/dev/null
rocprofv3 -L > gpurun_out/counters_$T.txt 2>&1 || true
for CFG in "200 20" "20 5"; do
  set -- $CFG; K=$1; W=$2; X=s${K}w${W}
  for S in PGS Newton; do
    A="--solver $S --steps $K --warmup $W --no-cpu-baseline --no-second-window"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_${S}_$X -- python bench.py $A > gpurun_out/bench_${T}_prof_${S}_$X.json 2>/dev/null
    head -4 $(find gpurun_out/prof_${T}_${S}_$X -name "*kernel_stats.csv" | head -1)
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_${T}_${S}_$X -- python bench.py $A > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_${T}_${S}_$X -- python bench.py $A > /dev/null 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq_${T}_${S}_$X -- python bench.py $A > /dev/null 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_flop_${T}_${S}_$X -- python bench.py $A > gpurun_out/pmc_flop_${T}_${S}_$X.log 2>&1 || echo "flop counter pass failed for $S $X"
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_${T}_${S}_$X -- python bench.py $A > gpurun_out/pmc_mfma_${T}_${S}_$X.log 2>&1 || echo "mfma counter pass failed for $S $X"
    echo "passes done: $S $X"
  done
done
python -c "import json; d=json.load(open('gpurun_out/bench_$T.json')); print(d['value'], d['ms_per_step'], d['health'], d['pgs']['value'], d['default_regime']['value'], d['pgs']['default_regime']['value'], d['cpu_baseline']['value'])"
python -c "import json; d=json.load(open('gpurun_out/bench_${T}_s20w5.json')); print(d['value'], d['ms_per_step'], d['pgs']['value'])"
