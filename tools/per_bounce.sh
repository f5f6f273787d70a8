#!/bin/bash
# usage (GPU box): bash tools/per_bounce.sh <workload> — per-bounce kernel durations (rocprofv3 kernel trace of ONE 64-spp batch) beside the
# 64-ray chunk counts of that batch ($PTAMD_DUMP_CHUNKS): does the rate hold up in the small launches of the late bounces?
set -u
WL=${1:-c3}; OUT=gpurun_out/perbounce_$WL; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PTAMD_DUMP_CHUNKS=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --no-cpu-baseline --no-kernel-pass --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ch = [l for l in open("$OUT/bench.err") if l.startswith("ptamd chunks")]
cc = [int(x) for x in ch[-2].split(":")[1].split()]; cs = [int(x) for x in ch[-1].split(":")[1].split()]
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "k_raygen" in n)   # the timed batch = the LAST raygen onwards
seq = rows[last:]
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
cl = [dur(r) for r in seq if "k_trace_closest" in r["Kernel_Name"]]
sh = [dur(r) for r in seq if "k_shade" in r["Kernel_Name"] and "records" not in r["Kernel_Name"]]
sd = [dur(r) for r in seq if "k_trace_shadow" in r["Kernel_Name"]]
print("$WL: bounce | closest chunks  ms  Grays/s | shade ms | shadow chunks  ms  Grays/s")
for b in range(len(cl)):
    print("%d | %8d %7.3f %6.2f | %7.3f | %8d %7.3f %6.2f" % (b, cc[b], cl[b], cc[b] * 64 / cl[b] / 1e6, sh[b] if b < len(sh) else 0, cs[b], sd[b] if b < len(sd) else 0, (cs[b] * 64 / sd[b] / 1e6) if b < len(sd) and sd[b] > 0 else 0))
t0 = int(seq[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seq)
busy = sum(dur(r) for r in seq)
print("batch wall %.3f ms, sum of kernel durations %.3f ms, gaps %.3f ms over %d launches" % ((t1 - t0) / 1e6, busy, (t1 - t0) / 1e6 - busy, len(seq)))
PY
