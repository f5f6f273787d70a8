# usage (GPU box): bash tools/sweep_spp.sh s1 s2 ... — C2/C3 with spp-per-step (= samples in flight) s and steps = 256/s
for s in "$@"; do
  for w in c2 c3; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --spp-per-step $s --steps $((256 / s)) > gpurun_out/ss_${s}_$w.json 2>gpurun_out/ss.err || exit 1
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ss_*.json")):
    d=json.load(open(f)); k=d["extra"]["kernel_ms"]
    print(f.split("ss_")[1][:-5].ljust(12), d["value"], "total kernel ms %.1f wall %.1f" % (sum(k.values()), d["extra"]["wall_ms"]), {a: round(b,1) for a,b in k.items()})
PY
