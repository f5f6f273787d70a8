# usage (GPU box): bash tools/r05_env_ab.sh <tag> "<name>:<ENV=val ENV2=val>" ... — C3 / C2 (/ C5 with R05_C5=1) kernel times under environment variants ("base:" = none)
tag=$1; shift
names=""
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; names="$names $name"
  for w in c3 c2 ${R05_C5:+c5}; do env $envs timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 6 > gpurun_out/${tag}_${name}_$w.json 2> gpurun_out/${tag}_${name}_$w.err || { tail -5 gpurun_out/${tag}_${name}_$w.err; exit 1; }; done
done
python - "$tag" $names <<'PY'
import json, sys, os
tag = sys.argv[1]
for v in sys.argv[2:]:
    for w in ("c3", "c2", "c5"):
        p = "gpurun_out/%s_%s_%s.json" % (tag, v, w)
        if not os.path.exists(p): continue
        d = json.load(open(p)); k = d["extra"]["kernel_ms"]; n = d["steps"]
        print(v, w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f" % (k["closest"] / n, k["shade"] / n, k["shadow"] / n, k["raygen"] / n), "mean", d["extra"]["mean_radiance"])
PY
