set -u
mkdir -p gpurun_out
for v in base v6 v6h v7h; do
  if [ $v != base ]; then export PTAMD_LIB=$GRAFT_REPO_ROOT/platinum_amd/csrc/libptamd_$v.so; fi
  for w in c3 c2; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 8 > gpurun_out/p3_${v}_$w.json 2> gpurun_out/p3_${v}_$w.err || exit 1; done
done
python - <<'PY'
import json
for v in ("base","v6","v6h","v7h"):
  for w in ("c3","c2"):
    d=json.load(open("gpurun_out/p3_%s_%s.json" % (v,w))); r=d["roofline_kernels"]["k_trace_closest"]; k=d["extra"]["kernel_ms"]; n=d["steps"]
    print(v, w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n))
PY
