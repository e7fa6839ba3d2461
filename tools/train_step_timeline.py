#!/usr/bin/env python3
"""One training step as a timeline from a rocprofv3 --kernel-trace csv (kernels AND the runtime's copy kernels, in start order):
start offset, duration, gap to the previous dispatch's end, name.  usage: train_step_timeline.py <dir> [step index from the end]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_adam")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = idx[-k - 1], idx[-k]
t0 = rows[a][0]
prev_end = rows[a][0]
busy = 0
for s, e, n in rows[a:b + 1]:
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {n[:80]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step {(rows[b][0] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
