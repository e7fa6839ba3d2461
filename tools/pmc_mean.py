#!/usr/bin/env python3
"""Mean of every collected counter per launch and kernel from one rocprofv3 --pmc pass:  pmc_mean.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = acc[row["Kernel_Name"]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if flt and flt not in k:
        continue
    print(k[:110], {c: (round(v[0] / v[1], 1), v[1]) for c, v in acc[k].items()})
