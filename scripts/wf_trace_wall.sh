#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_wall
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace -d "$OUT/trace" -o wall --output-format csv -- python3 "$ROOT/scripts/wf_wall.py" > "$OUT/run.log" 2>&1
grep pipeline "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_wf' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-40:]
t0 = int(last[0]['Start_Timestamp']); prev = t0
for r in last:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-14s start %9.1f dur %8.1f gap %8.1f grid %s lds %s" % (r['Kernel_Name'].split('::')[1].split('(')[0][:14], (st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3, r['Grid_Size_X'], r['LDS_Block_Size']))
    prev = en
PY
