#!/usr/bin/env python3
"""Mean FETCH_SIZE / WRITE_SIZE per launch and kernel from two rocprofv3 --pmc passes (tools/collect_profiles.sh).
rocprofv3 reports both counters in KB (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import sys
from collections import defaultdict


def mean_per_kernel(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


fetch, write = mean_per_kernel(sys.argv[1], "FETCH_SIZE"), mean_per_kernel(sys.argv[2], "WRITE_SIZE")
print("# mean per launch over the run (bench.py --steps 20 --warmup 2, config 2), two separate rocprofv3 --pmc passes; KB")
print("kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,launches")
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    print(f'"{k}",{fetch[k][0]:.1f},{write.get(k, (0.0, 0))[0]:.1f},{fetch[k][1]}')
