# usage (GPU box): bash tools/sweep_occ.sh — occupancy variants libptamd_w<waves>s<stack>p<pend>.so with matching PTAMD_BLOCKS_PER_CU
for lib in platinum_amd/csrc/libptamd_w*.so; do
  tag=$(basename $lib .so); w=${tag#libptamd_w}; w=${w%%s*}
  for wl in c2 c3; do
    PTAMD_LIB=$PWD/$lib PTAMD_BLOCKS_PER_CU=$w timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/oc_${tag}_$wl.json 2>gpurun_out/oc.err || { tail -3 gpurun_out/oc.err; exit 1; }
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/oc_*.json")):
    d=json.load(open(f)); r=d["roofline"]; k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(f.split("oc_")[1][:-5].ljust(28), d["value"], "closest/step %.2f shade/step %.2f shadow/step %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n))
PY
