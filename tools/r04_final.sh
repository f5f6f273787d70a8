# usage (GPU box): bash tools/r04_final.sh <tag> — GPU parity tests, the rocprofv3 passes for C3 and C2 (tools/profile_round.sh: they write
# profiles/<round>_pmc_<wl>.json, which the bench runs behind them read), default bench (C3), C2, C5 (through ingestion), the drop-in
# render(1) loop, the strong-scaling and two-rank rehearsals on one GPU (logical shards: labelled as rehearsals, no RCCL)
tag=${1:-r04a}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
bash tools/profile_round.sh c3 ${tag} > /dev/null 2>&1
bash tools/profile_round.sh c2 ${tag} > /dev/null 2>&1
bash tools/profile_round.sh c5 ${tag} > /dev/null 2>&1
timeout -k 10 400 python bench.py > gpurun_out/${tag}_c3.json 2> gpurun_out/${tag}_c3.err || { tail -5 gpurun_out/${tag}_c3.err; exit 1; }
timeout -k 10 300 python bench.py --workload c2 > gpurun_out/${tag}_c2.json 2> gpurun_out/${tag}_c2.err || { tail -5 gpurun_out/${tag}_c2.err; exit 1; }
timeout -k 10 300 python bench.py --workload c5 --steps 2 --no-cpu-baseline > gpurun_out/${tag}_c5.json 2> gpurun_out/${tag}_c5.err || tail -5 gpurun_out/${tag}_c5.err
timeout -k 10 300 python bench.py --workload c3 --gpus 2 --inproc --devices 0,0 --steps 4 --no-cpu-baseline > gpurun_out/${tag}_c3_inproc2.json 2> gpurun_out/${tag}_inproc.err || tail -5 gpurun_out/${tag}_inproc.err
timeout -k 10 300 python bench.py --workload c3 --steps 4 --no-cpu-baseline --drop-in-loop > gpurun_out/${tag}_c3_dropin.json 2> gpurun_out/${tag}_dropin.err || tail -5 gpurun_out/${tag}_dropin.err
timeout -k 10 300 python bench.py --workload c3 --gpus 2 --inproc --devices 0,0 --strong --spp 128 --no-cpu-baseline --no-kernel-pass > gpurun_out/${tag}_c3_strong2.json 2> gpurun_out/${tag}_strong.err || tail -5 gpurun_out/${tag}_strong.err
timeout -k 10 300 python bench.py --workload c3 --gpus 2 --rehearse-on-device0 --steps 2 --no-cpu-baseline --no-kernel-pass > gpurun_out/${tag}_c3_ranks2.json 2> gpurun_out/${tag}_ranks.err || tail -5 gpurun_out/${tag}_ranks.err
PTAMD_TWO_LEVEL=1 timeout -k 10 300 python bench.py --workload c3 --steps 4 --no-cpu-baseline > gpurun_out/${tag}_c3_two_level.json 2> gpurun_out/${tag}_tl.err || tail -3 gpurun_out/${tag}_tl.err
python - <<PY
import json
for w in ("c3","c2","c5","c3_dropin","c3_inproc2","c3_strong2","c3_ranks2","c3_two_level"):
    try: d=json.load(open("gpurun_out/${tag}_%s.json" % w))
    except Exception as e: print(w, "missing", e); continue
    k=d["extra"]["kernel_ms"]; n=d["steps"]; r=d["roofline"] or {}
    print(w, d["value"], "n_gpus", d["n_gpus"], d["scaling"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f acc %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n, k["accumulate"]/n),
          "dominant", r.get("kernel"), r.get("bound"), r.get("frac"), "binding", r.get("binding_secondary"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
mkdir -p gpurun_out/profiles_out; cp profiles/${tag}_* profiles/r04_pmc_* gpurun_out/profiles_out/ 2>/dev/null; cp gpurun_out/prof/*_bench_under_rocprof.json gpurun_out/profiles_out/ 2>/dev/null; ls gpurun_out/profiles_out | head -30
