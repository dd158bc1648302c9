#!/usr/bin/env python3
"""Device time of a workload's frames, kernel by kernel (rocprofv3 --kernel-trace over scripts/counter_frames.py; run on
a GPU box):  python scripts/kernel_sums.py <workload> <frames> [library]      (RTX_COUNTER_OPTS as in counter_frames.py)"""
import csv
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, frames = sys.argv[1], int(sys.argv[2])
d = tempfile.mkdtemp(prefix="rtx_ks_", dir="/tmp")
cmd = ["timeout", "-s", "KILL", "240", "rocprofv3", "--kernel-trace", "-d", d, "-o", "t", "--output-format", "csv", "--", sys.executable,
       os.path.join(ROOT, "scripts", "counter_frames.py"), workload, str(frames)] + sys.argv[3:4]
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=260)
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if r.returncode != 0 or not files:
    sys.exit("trace failed (%d): %s" % (r.returncode, r.stdout.decode(errors="replace")[-400:]))
per = {}
for f in files:
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_render|k_bin\w*|k_tile\w*|k_wf_\w+", row["Kernel_Name"])
        if m:
            e = per.setdefault(m.group(0), [0, 0.0])
            e[0] += 1
            e[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
shutil.rmtree(d, ignore_errors=True)
print("%s, per frame (%d frames): kernel, launches, ms" % (workload, frames))
for k in sorted(per):
    print("  %-18s %5d  %9.3f" % (k, per[k][0] // frames, per[k][1] / frames))
print("  %-18s        %9.3f" % ("all", sum(v[1] for v in per.values()) / frames))
