# usage (GPU box): bash tools/r05_sif.sh <tag> — throughput against samples in flight (spp per step) on C5 / C3 / C2
tag=$1
run() { w=$1; s=$2; k=$3; timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-kernel-pass --steps $k --warmup 1 --spp-per-step $s > gpurun_out/${tag}_${w}_$s.json 2> gpurun_out/${tag}_${w}_$s.err || { tail -3 gpurun_out/${tag}_${w}_$s.err; }; }
run c5 46 3; run c5 64 3; run c5 86 2; run c5 128 2
run c3 128 6; run c3 176 4; run c3 256 3
run c2 128 6; run c2 176 4; run c2 256 3
python - "$tag" <<'PY'
import json, sys, glob
tag = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/%s_c*_*.json" % tag)):
    try: d = json.load(open(f))
    except Exception: print(f, "failed"); continue
    print(f.split(tag + "_")[1][:-5].ljust(10), d["value"], "ms/step %.1f" % d["ms_per_step"], "in flight", d["config"]["samples_in_flight"], "batches/step", d["config"]["batches_per_step"])
PY
