# usage (on the GPU box): bash tools/sweep_variants.sh  — benches every platinum_amd/csrc/libptamd_*.so variant (C2, C3)
for lib in platinum_amd/csrc/libptamd_*.so; do
  tag=$(basename $lib .so)
  steps=4; case $tag in *wc) steps=1;; esac
  for w in c2 c3; do
    PTAMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps $steps > gpurun_out/sw_${tag}_$w.json 2>gpurun_out/sw.err || exit 1
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/sw_*.json")):
    d=json.load(open(f)); r=d["roofline"]; k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(f.split("sw_")[1][:-5].ljust(28), d["value"], "closest/step %.2f shade/step %.2f shadow/step %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n), "n/t", r["nodes_per_ray"], r["tris_per_ray"], "sh", r["shadow_kernel"]["nodes_per_ray"], r["shadow_kernel"]["tris_per_ray"])
PY
