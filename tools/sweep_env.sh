#!/bin/bash
# usage (GPU box): bash tools/sweep_env.sh VAR "v1 v2 ..." [workloads...] — benches the library under VAR=v for each value
VAR=$1; VALS=$2; shift 2; WLS=${@:-c3 c2}
for v in $VALS; do for w in $WLS; do
  env $VAR=$v timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 4 > gpurun_out/env_${VAR}_${v}_$w.json 2>gpurun_out/env_${VAR}_${v}_$w.err || { echo "$VAR=$v $w FAILED"; tail -3 gpurun_out/env_${VAR}_${v}_$w.err; }
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/env_${VAR}_*.json")):
    try: d=json.load(open(f))
    except Exception: continue
    k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(f.split("env_")[1][:-5].ljust(28), d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n))
PY
