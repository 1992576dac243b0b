#!/bin/bash
# GPU box: two SQ passes for several traversal modes.  usage: pmc_quick.sh "<modes>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for MODE in $1; do
  OUT=gpurun_out/pmcq_$MODE; mkdir -p $OUT
  CMD="python3 scripts/exp_trace.py --tris 5800000 --rays 33554432 --modes $MODE --blocks 6 --refill 36 --reps 2"
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/a -o x -- $CMD > $OUT/a.txt 2>&1
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -o x -- $CMD > $OUT/b.txt 2>&1
  grep "blocks" $OUT/a.txt | tail -1
  python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/x_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"].split("(")[0].replace("true, true","X"):
            acc[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    for k, v in acc.items():
        v = [x for x in v if x[1] > 5e6]
        if v: print(f"  mode $MODE {k:28s} {sum(x[0] for x in v)/len(v):.5g}   (n={len(v)})")
PY
done
