# usage (GPU box): bash tools/r02_baseline.sh <tag> — GPU parity tests, default bench (C3) + C2, then the rocprofv3 passes for C3
tag=${1:-r02a}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
timeout -k 10 300 python bench.py > gpurun_out/${tag}_c3.json 2> gpurun_out/${tag}_c3.err || { tail -5 gpurun_out/${tag}_c3.err; exit 1; }
timeout -k 10 300 python bench.py --workload c2 > gpurun_out/${tag}_c2.json 2> gpurun_out/${tag}_c2.err || { tail -5 gpurun_out/${tag}_c2.err; exit 1; }
python - <<PY
import json
for w in ("c3","c2"):
    d=json.load(open("gpurun_out/${tag}_%s.json" % w)); k=d["extra"]["kernel_ms"]; n=d["steps"]; r=d["roofline"]
    print(w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f acc %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n, k["accumulate"]/n),
          "dominant", r["kernel"], r["bound"], r["frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
bash tools/profile_round.sh c3 ${tag}
