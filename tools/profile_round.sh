#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh <workload> [tag] — the rocprofv3 passes behind profiles/<tag>_*:
#   1. kernel trace + stats of the DEFAULT bench command (per-kernel durations; must agree with bench.py's HIP-event times)
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE, 4. --pmc SQ instruction counters, 5. --pmc L1 / L2 request counters: separate passes of `bench.py --pmc-pass`
#      (full-size batches only), never combined with trace domains (MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots)
# then tools/summarize_prof.py (newest output of each pass) writes profiles/<tag>_<wl>_{summary.md,kernel_stats.csv} and
# profiles/<round>_pmc_<wl>.json.
set -u
WL=$1; TAG=${2:-r05}; OUT=gpurun_out/prof; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_trace -- python3 bench.py --workload $WL --no-cpu-baseline > $OUT/${WL}_bench_under_rocprof.json 2> $OUT/${WL}_trace.err || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${WL}_fetch -- python3 bench.py --workload $WL --pmc-pass --steps 2 > $OUT/${WL}_fetch.json 2> $OUT/${WL}_fetch.err || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${WL}_write -- python3 bench.py --workload $WL --pmc-pass --steps 2 > $OUT/${WL}_write.json 2> $OUT/${WL}_write.err || echo "write pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/${WL}_sq -- python3 bench.py --workload $WL --pmc-pass --steps 2 > $OUT/${WL}_sq.json 2> $OUT/${WL}_sq.err || echo "sq pass failed"
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${WL}_tc -- python3 bench.py --workload $WL --pmc-pass --steps 2 > $OUT/${WL}_tc.json 2> $OUT/${WL}_tc.err || echo "cache pass failed"
python3 tools/summarize_prof.py $OUT $WL $TAG > $OUT/${WL}_summary.txt 2>&1 || echo "summary failed"
tail -n 40 $OUT/${WL}_summary.txt
