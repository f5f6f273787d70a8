#!/bin/bash
# usage (GPU box): bash tools/pmc_icache.sh <workload> — instruction-cache counters per kernel (is k_shade's 78 KB of code a problem for the 64 KB
# instruction cache two CUs share?)
set -u
WL=${1:-c2}; OUT=gpurun_out/icache_$WL; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $OUT/pmc -- python3 bench.py --workload $WL --pmc-pass --steps 1 > $OUT/bench.json 2> $OUT/err.txt || { echo "pmc pass failed"; tail -5 $OUT/err.txt; }
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
if not f: raise SystemExit("no counter file")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f[0])):
    m = re.search(r"(k_\w+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    req, hit, miss = c.get("SQC_ICACHE_REQ", 0), c.get("SQC_ICACHE_HITS", 0), c.get("SQC_ICACHE_MISSES", 0)
    print("%-18s icache req %.3e hits %.3e misses %.3e miss rate %.4f | ifetch %.3e wave cycles %.3e" % (k, req, hit, miss, miss / req if req else 0, c.get("SQ_IFETCH", 0), c.get("SQ_WAVE_CYCLES", 0)))
PY
