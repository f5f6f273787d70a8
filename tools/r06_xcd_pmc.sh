# usage (GPU box): bash tools/r06_xcd_pmc.sh — L2 request / hit / miss counters of the trace kernels on C3, shipped claims against PT_XCD_CLAIMS (libptamd_xcd.so)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_xcd; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/default_tc -- python3 bench.py --workload c3 --pmc-pass --steps 2 > $OUT/default.json 2> $OUT/default.err || echo "default pass failed"
export PTAMD_LIB=$GRAFT_REPO_ROOT/platinum_amd/csrc/libptamd_xcd.so
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/xcd_tc -- python3 bench.py --workload c3 --pmc-pass --steps 2 > $OUT/xcd.json 2> $OUT/xcd.err || echo "xcd pass failed"
unset PTAMD_LIB
for v in default xcd; do echo "#### $v"; python3 tools/pmc_agg.py $OUT/${v}_tc k_trace; done > gpurun_out/r06_xcd_pmc.txt
python3 - <<'PY'
import json
for v in ("default", "xcd"):
    d = json.load(open("gpurun_out/prof_xcd/%s.json" % v)); print(v, "closest rays", d["extra"]["closest_rays"], "shadow rays", d["extra"]["shadow_rays"], "steps", d["steps"])
PY
cat gpurun_out/r06_xcd_pmc.txt
