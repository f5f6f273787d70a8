# usage (GPU box): bash tools/sweep_seg.sh — library variants x tiles-per-segment x tile order, C3 (+C2 for the contiguous orders)
run() { # lib tps strided workload
  tag=$(basename $1 .so)_t$2s$3_$4
  PTAMD_LIB=$PWD/$1 PTAMD_TILES_PER_SEG=$2 PTAMD_TILE_STRIDED=$3 timeout -k 10 200 python bench.py --workload $4 --no-cpu-baseline --steps 4 > gpurun_out/sg_$tag.json 2>gpurun_out/sg_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/sg_$tag.err; }
}
for lib in platinum_amd/csrc/libptamd.so platinum_amd/csrc/libptamd_nl3.so platinum_amd/csrc/libptamd_nl4.so; do
  for tps in 1 4; do for st in 0 1; do run $lib $tps $st c3; done; done
  run $lib 8 1 c3
  run $lib 4 1 c2
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/sg_*.json")):
    try: d=json.load(open(f))
    except Exception: continue
    k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(f.split("sg_")[1][:-5].ljust(30), d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n))
PY
