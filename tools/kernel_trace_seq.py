#!/usr/bin/env python3
"""Per-dispatch durations from a rocprofv3 --kernel-trace csv, in launch order:  kernel_trace_seq.py <dir> [name filter] [last N]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
rows.sort()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
last = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sel = [r for r in rows if flt in r[2]][-last:]
for a, b, n, g, w in sel:
    print(f"{(b - a) / 1e3:9.1f} us  grid {g:>8} wg {w:>4}  {n[:90]}")
