"""Prints the launches of the last frame of a rocprofv3 kernel trace (csv) of the level-by-level pipeline."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_wf' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    name = r['Kernel_Name'].split('::')[1].split('(')[0][:28]
    print("%-30s start %8.1f us  dur %8.1f us" % (name, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
