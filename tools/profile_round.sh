#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh <workload> — the rocprofv3 passes behind profiles/: kernel trace + stats, then
# FETCH_SIZE and WRITE_SIZE in their own --pmc passes (never combined with trace domains), all of the default bench command.
set -u
WL=$1; OUT=gpurun_out/prof; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_trace -- python3 bench.py --workload $WL --no-cpu-baseline > $OUT/${WL}_bench_under_rocprof.json 2> $OUT/${WL}_trace.err || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${WL}_fetch -- python3 bench.py --workload $WL --no-cpu-baseline --steps 1 --warmup 0 > $OUT/${WL}_fetch.json 2> $OUT/${WL}_fetch.err || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${WL}_write -- python3 bench.py --workload $WL --no-cpu-baseline --steps 1 --warmup 0 > $OUT/${WL}_write.json 2> $OUT/${WL}_write.err || echo "write pass failed"
ls $OUT/${WL}_trace/*/ | head
