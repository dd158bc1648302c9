#!/bin/bash
# Start and end of every render kernel of a workload's frames (rocprofv3 --kernel-trace), relative to the first:
#   bash scripts/kernel_times.sh [workload] [frames]      (run on a GPU box; output on stdout)
W=${1:-4k}; N=${2:-6}
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/rtx_kt_XXXX)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace -d $D -o t --output-format csv -- python3 $R/scripts/counter_frames.py $W $N > $D/log.txt 2>&1 || { tail -5 $D/log.txt; exit 1; }
python3 - $D <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "k_render" in r["Kernel_Name"] or "k_bin" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-40:]
    print("%-42s start %10.1f us  end %10.1f us  (%7.1f us)  grid %s" % (name, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?")))
PY
rm -rf $D
