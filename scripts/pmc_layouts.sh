#!/bin/bash
# GPU box: a few PMC passes over scripts/exp_treelets.py (2^26 rays, near-first walk) for one layout.
# usage: scripts/pmc_layouts.sh <tag> <layout> [extra args]
TAG=$1; LAYOUT=$2; shift; shift
OUT=gpurun_out/pmcl_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 scripts/exp_treelets.py --layouts $LAYOUT --modes 2 --anyhit 0 --reps 2 $@"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -o x -- $CMD > $OUT/p$i.txt 2>&1 || { echo "pass $i failed: $PMC"; tail -2 $OUT/p$i.txt; }
done
python3 scripts/pmc_sum.py $OUT k_trace_ | grep -v "true>" 
