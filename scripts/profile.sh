#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then separate PMC passes.
# usage (from the repository root, on the box): bash scripts/profile.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o "$TAG" --output-format csv -- python3 "$ROOT/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_under_trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/bench_under_trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass -d "$OUT/pmc_$name" -o "$TAG" --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -3 "$OUT/pmc_$name.log"; }
done
find "$OUT" -name "*.csv" | head -30
