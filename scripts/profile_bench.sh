#!/bin/bash
# Runs on the GPU box: kernel trace + PMC passes of bench.py, one rocprofv3 run each.
# usage: scripts/profile_bench.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --no-cpu-baseline --no-extra $@"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH --steps 5 --warmup 2 > $OUT/trace.json 2> $OUT/trace.err || { echo "trace pass failed"; tail -5 $OUT/trace.err; exit 1; }
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o bench -- $BENCH --steps 2 --warmup 1 > $OUT/pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i ($PMC) failed"; tail -3 $OUT/pmc$i.err; }
done
find $OUT -name "*.csv" | head -40
