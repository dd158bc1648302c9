#!/bin/bash
# Quick counter passes over the render kernel (bench.py, few steps).  usage: [BENCH_ARGS="--workload stress4k"] bash scripts/pmc_quick.sh <tag> ["counters of one pass" ...]
set -u
TAG=${1:-q}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" \
         "GRBM_GUI_ACTIVE"
fi
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass -d "$OUT/p$i" -o "$TAG" --output-format csv -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render<false" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k in sorted(acc):
    v = sorted(acc[k].values())
    print("%-32s %16.0f  (median of %d launches)" % (k, statistics.median(v), len(v)))
PY
