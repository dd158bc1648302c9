#!/usr/bin/env python3
"""Hardware counters of a workload's frames, kernel by kernel: scripts/counter_frames.py under `rocprofv3 --pmc`, one run
per group of counters (a group must fit the hardware: at most 4 TCC_, 4 TCP_, 8 SQ_ counters), summed per kernel name
and divided by the number of frames.

    python scripts/pmc_kernels.py <workload> <frames> <library or -> "COUNTER COUNTER ..." ["COUNTER ..." ...]"""
import csv
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    workload, frames, lib = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    env = dict(os.environ, TMPDIR="/tmp")
    base = tempfile.mkdtemp(prefix="rtx_pmc_", dir="/tmp")
    per, names = {}, []
    for n, group in enumerate(sys.argv[4:]):
        d = os.path.join(base, "p%d" % n)
        cmd = ["timeout", "-s", "KILL", "240", "rocprofv3", "--pmc"] + group.split() + ["-d", d, "-o", "c", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "scripts", "counter_frames.py"), workload, str(frames)] + ([] if lib == "-" else [lib])
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=260)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            sys.exit("pass %r failed (%d): %s" % (group, r.returncode, r.stdout.decode(errors="replace")[-400:]))
        names += group.split()
        for f in files:
            for row in csv.DictReader(open(f)):
                m = re.search(r"k_render|k_wf_\w+", row["Kernel_Name"])
                if m:
                    e = per.setdefault(m.group(0), {})
                    e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"]) / frames
    shutil.rmtree(base, ignore_errors=True)
    print("%s, per frame (%d frames)" % (workload, frames))
    for c in names:
        print("  %-28s" % c + "".join(" %s=%.4g" % (k, per[k].get(c, 0.0)) for k in sorted(per)))


if __name__ == "__main__":
    main()
