#!/bin/bash
# usage (GPU box): bash tools/bvh_timeline.sh <workload> — kernel-trace of one start + one step; prints the builder's timeline (the LAST build in the trace):
# per kernel name calls / total us, wall from k_flatten's start to k_reorder_tris' end, and the idle time between kernels.
set -u
WL=${1:-c3}; OUT=gpurun_out/bvhtl_$WL; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --no-cpu-baseline --no-kernel-pass --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    m = re.search(r"(k_\w+|rocclr_\w+|radix_sort\w*|merge\w*)", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:40]
names = [nm(r) for r in rows]
i0 = max(i for i, n in enumerate(names) if "k_flatten" in n)
i1 = max(i for i, n in enumerate(names) if "k_reorder_tris" in n)
seq = rows[i0:i1 + 1]
t0 = int(seq[0]["Start_Timestamp"]); t1 = int(seq[-1]["End_Timestamp"])
agg = collections.OrderedDict()
busy = 0
for r in seq:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = nm(r)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d; busy += d
gaps = sorted(((int(seq[i + 1]["Start_Timestamp"]) - int(seq[i]["End_Timestamp"])) / 1e3, nm(seq[i]), nm(seq[i + 1])) for i in range(len(seq) - 1))
print("$WL: build wall %.1f us, kernels busy %.1f us, %d launches" % ((t1 - t0) / 1e3, busy, len(seq)))
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("  %-62s %4d calls %9.1f us" % (k, c, d))
print("largest gaps (us):", [(round(g, 1), a, b) for g, a, b in gaps[-12:]])
print("sum of gaps %.1f us; gaps > 10 us: %d" % (sum(g for g, _, _ in gaps), sum(1 for g, _, _ in gaps if g > 10)))
PY
