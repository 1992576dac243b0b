#!/bin/bash
# GPU box: PMC passes over a short traversal-only run (scripts/exp_trace.py), one rocprofv3 run per pass.
# usage: scripts/pmc_exp.sh <tag> <mode> [tris]
TAG=$1; MODE=$2; TRIS=${3:-5800000}
OUT=gpurun_out/pmcx_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 scripts/exp_trace.py --tris $TRIS --rays 33554432 --modes $MODE --blocks 6 --refill 36 --reps 2"
i=0
for PMC in "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F16"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -o x -- $CMD > $OUT/p$i.txt 2>&1 || { echo "pass $i failed: $PMC"; tail -2 $OUT/p$i.txt; }
done
python3 - <<PY
import csv, glob, collections
vals = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/x_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
for k, v in vals.items():
    print(f"{k:45s} {v:.6g}")
PY
