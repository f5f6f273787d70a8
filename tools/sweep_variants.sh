# usage (on the GPU box): bash tools/sweep_variants.sh [workloads...] — benches the default library and every platinum_amd/csrc/libptamd_*.so variant
WLS=${@:-c3 c2}
for lib in platinum_amd/csrc/libptamd.so platinum_amd/csrc/libptamd_*.so; do
  [ -f $lib ] || continue
  tag=$(basename $lib .so)
  for w in $WLS; do
    PTAMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 4 > gpurun_out/sw_${tag}_$w.json 2>gpurun_out/sw_${tag}_$w.err || { echo "$tag $w FAILED"; tail -3 gpurun_out/sw_${tag}_$w.err; }
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/sw_*.json")):
    try: d=json.load(open(f))
    except Exception: continue
    k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(f.split("sw_")[1][:-5].ljust(28), d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n))
PY
