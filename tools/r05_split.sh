# usage (GPU box): bash tools/r05_split.sh <tag> — k_shade as one launch (baseline) against one launch per class group, class-0 kernel at 5 / 4 waves per SIMD
tag=$1
run() { # name env-assignments...
  name=$1; shift
  for w in c3 c2 ${R05_C5:+c5}; do env "$@" timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 6 > gpurun_out/${tag}_${name}_$w.json 2> gpurun_out/${tag}_${name}_$w.err || { tail -5 gpurun_out/${tag}_${name}_$w.err; exit 1; }; done
}
run base PTAMD_SHADE_SPLIT=0
run split5 PTAMD_SHADE_SPLIT=1
run split4 PTAMD_SHADE_SPLIT=1 PTAMD_LIB=$GRAFT_REPO_ROOT/platinum_amd/csrc/libptamd_s0w4.so
python - "$tag" <<'PY'
import json, sys, os
tag = sys.argv[1]
for v in ("base", "split5", "split4"):
    for w in ("c3", "c2", "c5"):
        p = "gpurun_out/%s_%s_%s.json" % (tag, v, w)
        if not os.path.exists(p): continue
        d = json.load(open(p)); k = d["extra"]["kernel_ms"]; n = d["steps"]
        print(v, w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f" % (k["closest"] / n, k["shade"] / n, k["shadow"] / n), "mean", d["extra"]["mean_radiance"])
PY
