#!/bin/bash
# The profiles kept under profiles/: for every bench workload, the rocprofv3 kernel statistics of `bench.py --workload W`
# and the bench line of a separate, unprofiled run (with its live counter passes and the CPU baseline).
# usage: bash scripts/profile_round.sh TAG [workloads]   (TAG names the files: r03 -> r03_4k_kernel_stats.csv, r03_bench_4k.json ...)
set -u
TAG=${1:?usage: profile_round.sh TAG [workloads]}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in ${@:-4k planes4k bunny4k 8k stress4k}; do
  STEPS=100; WARM=5; LIMIT=400; [ "$W" = stress4k ] && STEPS=5; [ "$W" = stress ] && { STEPS=3; WARM=1; LIMIT=900; }
  timeout -s KILL $LIMIT rocprofv3 --kernel-trace --stats -d "$OUT/trace_$W" -o "$W" --output-format csv -- python3 "$ROOT/bench.py" --workload $W --steps $STEPS --warmup $WARM --no-cpu-baseline --no-counters --no-extras > "$OUT/profiled_$W.json" 2> "$OUT/profiled_$W.err" || { echo "trace of $W failed"; tail -5 "$OUT/profiled_$W.err"; exit 1; }
  find "$OUT/trace_$W" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_${W}_kernel_stats.csv" \;
  rm -rf "$OUT/trace_$W"
  timeout -k 10 $LIMIT python3 "$ROOT/bench.py" --workload $W --steps $STEPS --warmup $WARM > "$OUT/${TAG}_bench_$W.json" 2> "$OUT/bench_$W.err" || { echo "bench of $W failed"; tail -5 "$OUT/bench_$W.err"; exit 1; }
  echo "== $W"; cut -d, -f1-6 "$OUT/${TAG}_${W}_kernel_stats.csv" | head -8
  python3 -c "
import json,sys
d=json.load(open('$OUT/${TAG}_bench_$W.json'))
r=d['roofline']
print(d['value'], d['unit'], d['ms_per_step'], 'ms; kernel', r['kernel_ms'], 'frac', r['frac'], 'traffic', r['traffic'], 'fast', (r.get('tolerance_build') or {}).get('kernel_ms'))"
done
