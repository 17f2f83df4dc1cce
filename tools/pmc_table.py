#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc run collected:
   python tools/pmc_table.py <rocprof output dir> [kernel substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short_name  # noqa: E402

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        a = acc[short_name(row["Kernel_Name"])][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
want = sys.argv[2:]
for k in sorted(acc):
    if want and not any(w in k for w in want):
        continue
    print(k, {c: round(v[1] / v[0], 1) for c, v in sorted(acc[k].items())}, "launches", max(v[0] for v in acc[k].values()))
