#!/usr/bin/env python3
"""HBM-side traffic of a workload's frames, kernel by kernel: scripts/counter_frames.py under `rocprofv3 --pmc`, one
pass for FETCH_SIZE and one for WRITE_SIZE (they do not fit one pass), summed per kernel name and divided by the
number of frames.  FETCH_SIZE counts kilobytes and tallies a 128-byte read request as 64 bytes on gfx950: doubled
here, as bench.py does; WRITE_SIZE (kilobytes) as it is.

    python scripts/traffic_by_kernel.py [workload] [frames] [library]

`variant` is passed on in RTX_VARIANT for builds that read it."""
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
    workload = sys.argv[1] if len(sys.argv) > 1 else "stress4k"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    env = dict(os.environ, TMPDIR="/tmp")
    base = tempfile.mkdtemp(prefix="rtx_traffic_", dir="/tmp")
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(base, counter)
        cmd = ["timeout", "-s", "KILL", "240", "rocprofv3", "--pmc", counter, "-d", d, "-o", "c", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "scripts", "counter_frames.py"), workload, str(frames)] + sys.argv[3:4]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=260)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            sys.exit("pass %s failed (%d): %s" % (counter, r.returncode, r.stdout.decode(errors="replace")[-400:]))
        for f in files:
            for row in csv.DictReader(open(f)):
                m = re.search(r"k_render|k_wf_\w+", row["Kernel_Name"])
                if not m:
                    continue
                e = per.setdefault(m.group(0), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
                e[row["Counter_Name"]] += float(row["Counter_Value"]) / frames
                if counter == "FETCH_SIZE":
                    e["launches"] += 1
    shutil.rmtree(base, ignore_errors=True)
    print("%s, per frame (%d frames): kernel, launches, fetched GB, written GB" % (workload, frames))
    for k in sorted(per):
        e = per[k]
        print("  %-14s %5d  %8.3f  %8.3f" % (k, e["launches"] // frames, 2 * e["FETCH_SIZE"] * 1024 / 1e9, e["WRITE_SIZE"] * 1024 / 1e9))
    print("  %-14s        %8.3f  %8.3f" % ("all", 2 * sum(e["FETCH_SIZE"] for e in per.values()) * 1024 / 1e9,
                                          sum(e["WRITE_SIZE"] for e in per.values()) * 1024 / 1e9))


if __name__ == "__main__":
    main()
