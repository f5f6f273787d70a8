#!/bin/bash
# usage: tools/pmc_sweep.sh <workload> <outdir>   (run on the GPU box; one rocprofv3 --pmc pass per counter group)
set -u
WL=$1; OUT=$2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py --workload $WL --pmc-pass --steps 2 > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i ($grp) failed"
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_INSTS_SALU
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY
GRPS
python3 tools/pmc_agg.py $OUT > $OUT/summary.txt 2>&1
