# usage (GPU box): bash tools/r05_variants.sh <tag> <variant...> — C3 / C2 (/ C5) kernel times of library variants built by tools/build_variant.sh ("default" = libptamd.so)
tag=$1; shift
for v in "$@"; do
  if [ $v = default ]; then unset PTAMD_LIB; else export PTAMD_LIB=$GRAFT_REPO_ROOT/platinum_amd/csrc/libptamd_$v.so; fi
  for w in c3 c2 ${R05_C5:+c5}; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 6 > gpurun_out/${tag}_${v}_$w.json 2> gpurun_out/${tag}_${v}_$w.err || { tail -5 gpurun_out/${tag}_${v}_$w.err; exit 1; }; done
done
python - "$tag" "$@" <<'PY'
import json, sys, os
tag = sys.argv[1]
for v in sys.argv[2:]:
    for w in ("c3", "c2", "c5"):
        p = "gpurun_out/%s_%s_%s.json" % (tag, v, w)
        if not os.path.exists(p): continue
        d = json.load(open(p)); k = d["extra"]["kernel_ms"]; n = d["steps"]
        print(v, w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f" % (k["closest"] / n, k["shade"] / n, k["shadow"] / n))
PY
