# usage (GPU box): bash tools/r05_quick.sh <tag> [tests] — optional GPU suite, then C3 / C2 / C5 bench lines (8 steps, no CPU leg) and one summary line each
tag=${1:-q5}
if [ "${2:-}" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log; grep "full size" gpurun_out/${tag}_tests.log
fi
for w in c3 c2; do timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 8 > gpurun_out/${tag}_$w.json 2>gpurun_out/${tag}_$w.err || { tail -5 gpurun_out/${tag}_$w.err; exit 1; }; done
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 3 > gpurun_out/${tag}_c5.json 2>gpurun_out/${tag}_c5.err || { tail -5 gpurun_out/${tag}_c5.err; exit 1; }
python - <<PY
import json
for w in ("c3","c2","c5"):
    d=json.load(open("gpurun_out/${tag}_%s.json" % w)); r=d["roofline"]; k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f acc %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n, k["accumulate"]/n), "spp/step", d["config"]["spp_per_step"])
PY
