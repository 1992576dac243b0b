#!/bin/bash
# GPU box: vector-ALU counters of the traversal launches of a python script of this repository (one rocprofv3 --pmc pass).
# usage: scripts/pmc_valu.sh <tag> <kernel name filter> <script and its args...>
TAG=$1; FLT=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o x -- python3 "$@" > $OUT/p1.txt 2>&1 || { echo "pass failed"; tail -2 $OUT/p1.txt; exit 1; }
python3 scripts/pmc_sum.py $OUT "$FLT"
find $OUT -name "*.csv" -size +1M -delete
