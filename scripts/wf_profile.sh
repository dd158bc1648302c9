#!/bin/bash
# Kernel trace of the level-by-level pipeline on one workload.  usage: bash scripts/wf_profile.sh <tag> <workload>
set -u
TAG=${1:-wf}; WL=${2:-4k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o "$TAG" --output-format csv -- python3 "$ROOT/scripts/pipelines.py" $WL > "$OUT/run.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/run.log"; exit 1; }
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats.csv" \;
cat "$OUT/run.log" | grep pipeline
cut -d, -f1-8 "$OUT/${TAG}_kernel_stats.csv" | head -30
