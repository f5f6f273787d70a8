# usage (GPU box): bash tools/r02_final.sh <tag> — GPU parity tests, default bench (C3), C2, C5 (through ingestion), the rocprofv3 passes for
# C3 and C2, the FETCH_SIZE pass of the two-level structure on C3, and the in-process two-logical-shard rehearsal
tag=${1:-r02c}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
bash tools/profile_round.sh c3 ${tag} > /dev/null 2>&1
bash tools/profile_round.sh c2 ${tag} > /dev/null 2>&1
timeout -k 10 400 python bench.py > gpurun_out/${tag}_c3.json 2> gpurun_out/${tag}_c3.err || { tail -5 gpurun_out/${tag}_c3.err; exit 1; }
timeout -k 10 300 python bench.py --workload c2 > gpurun_out/${tag}_c2.json 2> gpurun_out/${tag}_c2.err || { tail -5 gpurun_out/${tag}_c2.err; exit 1; }
timeout -k 10 300 python bench.py --workload c5 --steps 2 --no-cpu-baseline > gpurun_out/${tag}_c5.json 2> gpurun_out/${tag}_c5.err || tail -5 gpurun_out/${tag}_c5.err
timeout -k 10 300 python bench.py --workload c3 --gpus 2 --inproc --devices 0,0 --steps 4 --no-cpu-baseline > gpurun_out/${tag}_c3_inproc2.json 2> gpurun_out/${tag}_inproc.err || tail -5 gpurun_out/${tag}_inproc.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PTAMD_TWO_LEVEL=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/c3tl_fetch -- python3 bench.py --workload c3 --pmc-pass --steps 2 > gpurun_out/prof/c3tl_fetch.json 2> gpurun_out/prof/c3tl_fetch.err || echo "two-level fetch pass failed"
PTAMD_TWO_LEVEL=1 timeout -k 10 300 python bench.py --workload c3 --steps 4 --no-cpu-baseline > gpurun_out/${tag}_c3_two_level.json 2> gpurun_out/${tag}_tl.err || tail -3 gpurun_out/${tag}_tl.err
python - <<PY
import json
for w in ("c3","c2","c5","c3_inproc2","c3_two_level"):
    try: d=json.load(open("gpurun_out/${tag}_%s.json" % w))
    except Exception as e: print(w, "missing", e); continue
    k=d["extra"]["kernel_ms"]; n=d["steps"]; r=d["roofline"]
    print(w, d["value"], "ms/step %.2f" % d["ms_per_step"], "closest %.2f shade %.2f shadow %.2f raygen %.2f acc %.2f" % (k["closest"]/n, k["shade"]/n, k["shadow"]/n, k["raygen"]/n, k["accumulate"]/n),
          "dominant", r["kernel"], r["bound"], r["frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
