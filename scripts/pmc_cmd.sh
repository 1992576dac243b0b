#!/bin/bash
# GPU box: a few PMC passes (one rocprofv3 run each) over a python script of this repository.
# usage: scripts/pmc_cmd.sh <tag> <kernel name filter> <script and its args...>
TAG=$1; FLT=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -o x -- python3 "$@" > $OUT/p$i.txt 2>&1 || { echo "pass $i failed: $PMC"; tail -2 $OUT/p$i.txt; exit 1; }
done
python3 scripts/pmc_sum.py $OUT "$FLT"
find $OUT -name "*.csv" -size +1M -delete
