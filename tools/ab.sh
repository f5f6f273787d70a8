# usage (GPU box): bash tools/ab.sh <tag> <workloads: "c3 c2"> <variant[:ENV=VAL,...]> ... — kernel times of library variants (tools/build_variant.sh; "default" = libptamd.so)
tag=$1; wls=$2; shift 2
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  ( if [ $v != default ]; then export PTAMD_LIB=$GRAFT_REPO_ROOT/platinum_amd/csrc/libptamd_$v.so; fi
    for kv in ${envs//,/ }; do export $kv; done
    name=${spec//[:=,]/_}
    for w in $wls; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 6 > gpurun_out/${tag}_${name}_$w.json 2> gpurun_out/${tag}_${name}_$w.err || { tail -5 gpurun_out/${tag}_${name}_$w.err; exit 1; }; done ) || exit 1
done
python - "$tag" "$wls" "$@" <<'PY'
import json, sys, os, re
tag, wls = sys.argv[1], sys.argv[2].split()
for spec in sys.argv[3:]:
    name = re.sub(r"[:=,]", "_", spec)
    for w in wls:
        p = "gpurun_out/%s_%s_%s.json" % (tag, name, w)
        if not os.path.exists(p): continue
        d = json.load(open(p)); k = d["extra"]["kernel_ms"]; n = d["steps"]
        print("%-28s %-5s %9.1f  ms/step %.2f  closest %.2f shade %.2f shadow %.2f raygen %.2f" % (spec, w, d["value"], d["ms_per_step"], k["closest"] / n, k["shade"] / n, k["shadow"] / n, k["raygen"] / n))
PY
