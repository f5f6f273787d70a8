#!/bin/bash
# usage (GPU box): PTAMD_LIB=... bash tools/pmc_tcp.sh <workload> <outdir> — is a kernel bound by the vector L1 (TA / TCP)?  One rocprofv3 --pmc pass per group.
set -u
WL=$1; OUT=$2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py --workload $WL --pmc-pass --steps 2 > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i ($grp) failed"
done <<'GRPS'
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
GRPS
python3 tools/pmc_agg.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | grep -A40 "k_trace_closest<false, false>" | head -45
