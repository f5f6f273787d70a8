#!/bin/bash
# usage (GPU box): bash tools/calib_sweep.sh — the record-gather calibration behind profiles/r03_calib_gather.md
# (tools/calib_gather2.hip: access shape x table size x records in flight), then counter passes for the shapes that matter.
set -u
OUT=gpurun_out/calib; mkdir -p $OUT
B=tools/_build/calib_gather2
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: > $OUT/sweep.jsonl
for mode in 0 1 2 3 4 5 6 7; do
  for mib in 2 15 32 98 1024; do
    for infl in 2 4 8 16; do
      timeout -k 5 60 $B $mode $mib 64 $infl >> $OUT/sweep.jsonl 2>> $OUT/sweep.err || echo "{\"mode\": $mode, \"table_MiB\": $mib, \"inflight\": $infl, \"failed\": true}" >> $OUT/sweep.jsonl
    done
  done
  echo "mode $mode done"
done
# counters: what one record costs the vector L1 (tag accesses, L2 requests, TA busy cycles) per shape, 32 MiB table (the C3 node array's size)
for mode in 0 1 2 3 5; do
  timeout -k 10 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp_m$mode -- $B $mode 32 64 4 > $OUT/pmc_tcp_m$mode.json 2> $OUT/pmc_tcp_m$mode.err || echo "tcp pass $mode failed"
  timeout -k 10 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta_m$mode -- $B $mode 32 64 4 > $OUT/pmc_ta_m$mode.json 2> $OUT/pmc_ta_m$mode.err || echo "ta pass $mode failed"
  timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/pmc_tcc_m$mode -- $B $mode 32 64 4 > $OUT/pmc_tcc_m$mode.json 2> $OUT/pmc_tcc_m$mode.err || echo "tcc pass $mode failed"
done
python3 tools/calib_report.py $OUT > $OUT/report.md 2> $OUT/report.err || echo "report failed"
tail -n 80 $OUT/report.md
