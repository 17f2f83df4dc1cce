#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 kernel_stats.csv (first one found under the directory given): calls, total ms, average us."""
import csv, glob, os, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = d if os.path.isfile(d) else sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
for r in list(csv.DictReader(open(f)))[:top]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
