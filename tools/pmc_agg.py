#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection.csv files: mean counter value per dispatch, per (short) kernel name."""
import csv, glob, sys
from collections import defaultdict
def short(name):
    for k in ("k_trace_closest<false, false>", "k_trace_shadow<false, false>", "k_trace_closest<false, true>", "k_trace_shadow<false, true>", "k_trace_closest<(bool)0, (bool)0>", "k_trace_closest<(bool)0, (bool)1>", "k_trace_shadow<(bool)0, (bool)0>", "k_trace_shadow<(bool)0, (bool)1>",
              "k_shade", "k_raygen", "k_accumulate"):
        if k in name: return k
    return None
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in acc:
    print("==", k)
    for c, (n, v) in sorted(acc[k].items()):
        print(f"  {c:40s} n={n:4d} mean/dispatch={v/n:16.1f} total={v:18.1f}")
