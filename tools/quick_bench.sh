# usage (GPU box): bash tools/quick_bench.sh [tag] — GPU parity tests, then C2 + C3 bench (16 steps), one summary line each
tag=${1:-qb}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -2 gpurun_out/${tag}_tests.log
for w in c2 c3; do timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 16 > gpurun_out/${tag}_$w.json 2>gpurun_out/${tag}.err || exit 1; done
python - <<PY
import json
for w in ("c2","c3"):
    d=json.load(open("gpurun_out/${tag}_%s.json" % w)); r=d["roofline"]; k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n), "n/t", r["nodes_per_ray"], r["tris_per_ray"], "frac", r["frac"])
PY
