#!/usr/bin/env python3
"""Gaps between the kernels of consecutive denoising steps from a rocprofv3 --kernel-trace csv:  step_gaps.py <dir> [last N steps]
(start of a kernel - end of the previous one on the device's clock; a step = from one k_n16_edge launch to the next)."""
import csv
import glob
import statistics
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
idx = [i for i, r in enumerate(rows) if "k_n16_edge" in r[2]]
idx = idx[-n_last - 1:]
gaps, durs, steps = {}, {}, []
for a, b in zip(idx[:-1], idx[1:]):
    seq = rows[a:b + 1]
    if len(seq) > 8:
        continue                                         # (not a plain step: a trajectory start, a bind)
    steps.append((seq[-1][0] - seq[0][0]) / 1e3)
    for x, y in zip(seq[:-1], seq[1:]):
        k = x[2][x[2].find("k_"):][:18] if "k_" in x[2] else x[2][:18]
        k2 = y[2][y[2].find("k_"):][:18] if "k_" in y[2] else y[2][:18]
        gaps.setdefault(f"{k} -> {k2}", []).append((y[0] - x[1]) / 1e3)
        durs.setdefault(k, []).append((x[1] - x[0]) / 1e3)
print(f"{len(steps)} steps: median step {statistics.median(steps):.2f} us")
for k, v in durs.items():
    print(f"  kernel {k:20s} median {statistics.median(v):6.2f} us")
for k, v in gaps.items():
    print(f"  gap    {k:42s} median {statistics.median(v):6.2f} us")
