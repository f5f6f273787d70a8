#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection.csv files under a directory: mean counter value per dispatch per kernel (template arguments
kept, parameter lists dropped).  usage: pmc_agg2.py <dir> [kernel-substring]"""
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void pt::", "")).strip()
        if flt and flt not in k:
            continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(acc):
    print("==", k)
    for c, (n, v) in sorted(acc[k].items()):
        print(f"  {c:40s} n={n:4d} mean/dispatch={v/n:16.1f} total={v:18.1f}")
