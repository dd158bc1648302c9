#!/bin/bash
# Counter passes over the kernels of one workload of scripts/pipelines.py.  usage: bash scripts/pmc_wf.sh <tag> <workload> ["counters of one pass" ...]
set -u
TAG=${1:-q}; WL=${2:-4k}
shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"
fi
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass -d "$OUT/p$i" -o "$TAG" --output-format csv -- python3 "$ROOT/scripts/pipelines.py" $WL > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics, re
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_wf_\w+|k_render)", r["Kernel_Name"])
        if not m: continue
        key = (m.group(1), r["Counter_Name"])
        acc.setdefault(key, {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k in sorted(acc):
    v = list(acc[k].values())
    print("%-14s %-24s sum/launches %16.0f  launches %4d  max %14.0f" % (k[0], k[1], sum(v) / len(v), len(v), max(v)))
PY
