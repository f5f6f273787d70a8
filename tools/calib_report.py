#!/usr/bin/env python3
"""Markdown report of tools/calib_sweep.sh: records/s by access shape, table size and records in flight; counters per record."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
NAMES = {0: "lane-private 64 B (4 x dwordx4 per lane)", 1: "quad-cooperative 64 B (4 lanes x 16 B, one instruction)",
         2: "oct-cooperative 128 B (8 lanes x 16 B, one instruction)", 3: "lane-private 128 B (8 x dwordx4 per lane)",
         4: "lane-private 16 B (one dwordx4)", 5: "quad-cooperative 64 B by LDS-DMA + 4 x ds_read_b128 per lane",
         6: "lane-private 4 B (one dword)", 7: "pair-cooperative 32 B (2 lanes x 16 B)"}
rows = [json.loads(l) for l in open(os.path.join(out, "sweep.jsonl")) if l.startswith("{")]
best = defaultdict(dict)
for r in rows:
    if r.get("failed"):
        continue
    k = (r["mode"], r["table_MiB"])
    if r["inflight"] not in best[k] or best[k][r["inflight"]]["Grecords_per_s"] < r["Grecords_per_s"]:
        best[k][r["inflight"]] = r
print("## Records per second (G/s) by access shape, table size and independent records in flight\n")
sizes = sorted({r["table_MiB"] for r in rows if not r.get("failed")})
for mode in sorted(NAMES):
    if not any(k[0] == mode for k in best):
        continue
    print("**mode %d — %s**\n" % (mode, NAMES[mode]))
    infls = sorted({i for k, v in best.items() if k[0] == mode for i in v})
    print("| table MiB | " + " | ".join("%d in flight" % i for i in infls) + " | best GB/s | lane-addresses / clk / CU (2.1 GHz) |")
    print("|---|" + "---|" * (len(infls) + 2))
    for s in sizes:
        v = best.get((mode, s), {})
        if not v:
            continue
        top = max(v.values(), key=lambda r: r["Grecords_per_s"])
        print("| %g | " % s + " | ".join("%.1f" % v[i]["Grecords_per_s"] if i in v else "-" for i in infls) +
              " | %.0f | %.2f |" % (top["GBs_fetched"], top["lane_addresses_per_clk_per_CU_at_2.1GHz"]))
    print()
print("## Counters per record (32 MiB table, 4 in flight)\n")
print("| mode | records | kernel ms | TCP tag accesses / record | TCP->TCC read requests / record | TCP_TOTAL_ACCESSES / record | TA busy cycles / record (sum over TAs) | TCC hit rate | TCC_REQ / record |")
print("|---|---|---|---|---|---|---|---|---|")
for mode in sorted(NAMES):
    vals = {}
    recs = None
    for kind in ("tcp", "ta", "tcc"):
        j = os.path.join(out, "pmc_%s_m%d.json" % (kind, mode))
        if os.path.exists(j):
            try:
                recs = json.loads([l for l in open(j) if l.startswith("{")][-1])["records"]
            except Exception:
                pass
        f = glob.glob(os.path.join(out, "pmc_%s_m%d" % (kind, mode), "**", "*_counter_collection.csv"), recursive=True)
        if not f:
            continue
        per = defaultdict(list)
        for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
            if "k_gather" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in per.items():
            vals[c] = v[-1]  # the last dispatch = a timed repetition (full size)
    if not vals or not recs:
        continue
    g = lambda c: vals.get(c, float("nan"))
    hit = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    print("| %d | %.0f | - | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (mode, recs, g("TCP_TOTAL_CACHE_ACCESSES_sum") / recs, g("TCP_TCC_READ_REQ_sum") / recs,
          g("TCP_TOTAL_ACCESSES_sum") / recs, g("TA_TA_BUSY_sum") / recs, hit, g("TCC_REQ_sum") / recs))
