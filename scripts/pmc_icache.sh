#!/bin/bash
# Instruction-cache counters of the render kernel with and without quad mode.  usage: bash scripts/pmc_icache.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_icache
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for opts in "" "8=0" "8=0,7=0"; do
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY"; do
    i=$((i+1))
    RTX_COUNTER_OPTS="$opts" timeout -s KILL 120 rocprofv3 --pmc $pass -d "$OUT/p$i" -o c --output-format csv -- python3 "$ROOT/scripts/counter_frames.py" 4k 8 > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; }
    echo "== options '$opts' =="
    python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / 8
for k in sorted(acc): print("  %-28s %16.0f" % (k, acc[k]))
PY
  done
done
