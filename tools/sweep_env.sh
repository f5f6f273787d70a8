# usage (GPU box): bash tools/sweep_env.sh VAR v1 v2 ...  — benches C2 and C3 with VAR set to each value
var=$1; shift
for v in "$@"; do
  for w in c2 c3; do
    env $var=$v timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 16 > gpurun_out/se_${var}_${v}_$w.json 2>gpurun_out/se.err || exit 1
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/se_${var}_*.json")):
    d=json.load(open(f)); r=d["roofline"]; k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(f.split("se_")[1][:-5].ljust(28), d["value"], "closest/step %.2f shade/step %.2f shadow/step %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n), "n/t", r["nodes_per_ray"], r["tris_per_ray"])
PY
