"""Mean counter values per kernel over the large dispatches of a rocprofv3 --pmc run (those that ran at least PMC_SUM_FRACTION,
default 0.5, of the longest one's time).  usage: pmc_sum.py <dir> [name filter]"""
import collections
import csv
import glob
import os
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append((float(r["Counter_Value"]), dur))
    for (k, c), v in sorted(acc.items()):
        big = [x for x in v if x[1] > float(os.environ.get("PMC_SUM_FRACTION", "0.5")) * max(y[1] for y in v)]
        print(f"{k:62s} {c:32s} {sum(x[0] for x in big) / len(big):.6g}  (n={len(big)}, {sum(x[1] for x in big) / len(big) / 1e6:.2f} ms)")
