# usage (GPU box): bash tools/r06_final.sh <tag> — GPU suite, the rocprofv3 passes for C3 / C2 / C5 (tools/profile_round.sh; their logs are KEPT and a failed
# pass or a stale counter file stops the script: ADVICE r4), then the bench lines: default (C3), C2, C5, the drop-in render(1) loop, logical-shard
# rehearsals on one GPU (labelled REHEARSAL, no RCCL), two-level
# stage (2nd argument): "profiles" = suite + rocprofv3 passes only (copy gpurun_out/profiles_out/* into profiles/ afterwards: only gpurun_out/ comes
# back from the box), "bench" = the bench lines only (reads the committed profiles/r06_pmc_*.json), default = both in one call
tag=${1:-r06}; stage=${2:-all}
fail() { echo "r06_final: $*"; exit 1; }
if [ "$stage" != bench ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log; grep "full size" gpurun_out/${tag}_tests.log
for w in c3 c2 c5 c3xl; do
  bash tools/profile_round.sh $w ${tag} > gpurun_out/${tag}_profile_$w.log 2>&1
  grep -n "pass failed\|summary failed\|summarize_prof.py:" gpurun_out/${tag}_profile_$w.log && fail "profile pass of $w failed (gpurun_out/${tag}_profile_$w.log)"
done
mkdir -p gpurun_out/profiles_out; cp profiles/${tag}_* profiles/r06_pmc_* gpurun_out/profiles_out/ 2>/dev/null; cp gpurun_out/prof/*_bench_under_rocprof.json gpurun_out/profiles_out/ 2>/dev/null; ls gpurun_out/profiles_out | head -30
fi
[ "$stage" = profiles ] && exit 0
bench() { # name timeout args...
  n=$1; t=$2; shift 2
  timeout -k 10 $t python bench.py "$@" > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err || { tail -8 gpurun_out/${tag}_$n.err; fail "bench $n failed"; }
  grep -q "STALE" gpurun_out/${tag}_$n.err && fail "bench $n used a STALE counter file (the profile passes above did not update profiles/)"
}
bench c3 400 --steps 20 --warmup 5
bench c2 300 --workload c2
bench c5 300 --workload c5 --steps 3 --no-cpu-baseline
bench c3xl 300 --workload c3xl --steps 4 --no-cpu-baseline
bench c3_dropin 300 --workload c3 --steps 4 --no-cpu-baseline --drop-in-loop
bench c3_inproc2 300 --workload c3 --gpus 2 --inproc --devices 0,0 --steps 4 --no-cpu-baseline
bench c3_strong2 300 --workload c3 --gpus 2 --inproc --devices 0,0 --strong --spp 128 --no-cpu-baseline --no-kernel-pass
bench c3_ranks2 300 --workload c3 --gpus 2 --rehearse-on-device0 --steps 2 --no-cpu-baseline --no-kernel-pass
PTAMD_TWO_LEVEL=1 bench c3_two_level 300 --workload c3 --steps 4 --no-cpu-baseline
python - <<PY
import json
for w in ("c3","c2","c5","c3xl","c3_dropin","c3_inproc2","c3_strong2","c3_ranks2","c3_two_level"):
    try: d=json.load(open("gpurun_out/${tag}_%s.json" % w))
    except Exception as e: print(w, "missing", e); continue
    k=d["extra"]["kernel_ms"]; n=d["steps"]; r=d["roofline"] or {}
    print(w, d["value"], "n_gpus", d["n_gpus"], d["scaling"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f acc %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n, k["accumulate"]/n),
          "dominant", r.get("kernel"), r.get("bound"), r.get("frac"), "binding", r.get("binding_secondary"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
mkdir -p gpurun_out/profiles_out; cp profiles/${tag}_* profiles/r06_pmc_* gpurun_out/profiles_out/ 2>/dev/null; cp gpurun_out/prof/*_bench_under_rocprof.json gpurun_out/profiles_out/ 2>/dev/null; ls gpurun_out/profiles_out | head -30
