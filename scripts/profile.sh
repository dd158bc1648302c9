#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then separate counter passes.
# usage (from the repository root, on the box): bash scripts/profile.sh <tag>
# Writes gpurun_out/prof_<tag>/{<tag>_kernel_stats.csv, <tag>_pmc_k_render.json}; copy those into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o "$TAG" --output-format csv -- python3 "$ROOT/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_under_trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/bench_under_trace.log"; exit 1; }
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass -d "$OUT/pmc_$i" -o "$TAG" --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_$i.log" 2>&1 || { echo "pmc pass $i ($pass) failed"; tail -3 "$OUT/pmc_$i.log"; }
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, shutil, statistics, sys
out, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, tag + "_kernel_stats.csv"))
acc = {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render<false" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
summary = {}
for k, per_dispatch in sorted(acc.items()):
    v = sorted(per_dispatch.values())
    summary[k] = {"launches": len(v), "median": statistics.median(v), "min": v[0], "max": v[-1]}
json.dump(summary, open(os.path.join(out, tag + "_pmc_k_render.json"), "w"), indent=1)
for k, v in summary.items():
    print("%-32s %16.0f (%d launches)" % (k, v["median"], v["launches"]))
PY
grep -i "k_render" "$OUT/${TAG}_kernel_stats.csv" | head -3
