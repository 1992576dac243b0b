#!/bin/bash
# GPU box: PMC passes over the wavefront path tracer (scripts/exp_render.py), one rocprofv3 run per pass.
# usage: scripts/pmc_render.sh <tag> [exp_render args]
set -o pipefail
TAG=${1:-render}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
i=0
for PMC in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o render -- python3 scripts/exp_render.py --frames 12 "$@" > $OUT/pmc$i.txt 2> $OUT/pmc$i.err || { echo "pmc pass $i ($PMC) failed"; tail -3 $OUT/pmc$i.err; exit 1; }
  grep spp $OUT/pmc$i.txt
done
find $OUT -name "*counter_collection.csv"
