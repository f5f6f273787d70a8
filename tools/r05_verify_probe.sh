# usage (GPU box): bash tools/r05_verify_probe.sh — experiment builds that verify every closest-hit winner against its own box inside k_shade: cost and failure counts
for v in default vh244 vh615; do
  if [ $v = default ]; then unset PTAMD_LIB; else export PTAMD_LIB=$GRAFT_REPO_ROOT/platinum_amd/csrc/libptamd_$v.so; fi
  for w in c3 c2; do
    PTAMD_DUMP_CHUNKS=1 timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 6 > gpurun_out/vp_${v}_$w.json 2> gpurun_out/vp_${v}_$w.err || { tail -3 gpurun_out/vp_${v}_$w.err; exit 1; }
    echo "$v $w: $(grep 'fail their own box' gpurun_out/vp_${v}_$w.err | sort | uniq -c | sort -rn | head -4 | tr '\n' ';')"
  done
done
python - <<'PY'
import json
for v in ("default", "vh244", "vh615"):
    for w in ("c3", "c2"):
        d = json.load(open("gpurun_out/vp_%s_%s.json" % (v, w))); k = d["extra"]["kernel_ms"]; n = d["steps"]
        print(v, w, d["value"], "closest %.2f shade %.2f shadow %.2f" % (k["closest"] / n, k["shade"] / n, k["shadow"] / n))
PY
