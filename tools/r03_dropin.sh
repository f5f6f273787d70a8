#!/bin/bash
# usage (GPU box): bash tools/r03_dropin.sh — the reference frontend's call pattern (render() = ONE sample per call):
# throughput by samples in flight per batch (no merging possible: one call per batch, then wait) against the drop-in loop
# (render(1) called back to back, merged by the library), C3 and C2.  Writes gpurun_out/dropin_*.json; summary on stdout.
for w in c3 c2; do
  for s in 1 4 16 64; do
    k=$((256 / s)); [ $k -gt 64 ] && k=64
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-kernel-pass --spp-per-step $s --steps $k --warmup 1 > gpurun_out/dropin_${w}_sif$s.json 2> gpurun_out/dropin_${w}_sif$s.err || echo "$w sif $s failed"
  done
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-kernel-pass --spp-per-step 64 --steps 4 --warmup 1 --drop-in-loop > gpurun_out/dropin_${w}_loop.json 2> gpurun_out/dropin_${w}_loop.err || echo "$w loop failed"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/dropin_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable"); continue
    c = d["config"]
    print(f.split("dropin_")[1][:-5].ljust(12), "%9.1f Msamples/s" % d["value"], "spp/step", c["spp_per_step"], "steps", d["steps"], "batches", c.get("batches"), "|", c.get("call_pattern"))
PY
