"""Summarises a gpurun_out/prof_<tag> directory (made by scripts/profile_bench.sh) into profiles/<tag>_summary.md."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
out = f"profiles/{tag}_summary.md"
lines = [f"# rocprofv3 summary `{tag}`", "", "Command: `bench.py --no-cpu-baseline --no-extra` "
         "(kernel trace: --steps 5 --warmup 2; PMC passes: --steps 2 --warmup 1, one rocprofv3 run per pass).", ""]
ks = os.path.join(src, "trace", "bench_kernel_stats.csv")
lines += ["## Kernel trace (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
for r in csv.DictReader(open(ks)):
    lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.3f} | {r['Percentage']} |")
try:
    bj = json.loads(open(os.path.join(src, "trace.json")).read().strip().splitlines()[-1])
    lines += ["", f"bench line under the profiler: value {bj['value']} {bj['unit']}, kernel_ms {bj['roofline']['kernel_ms']}, "
              f"bytes/ray {bj['roofline']['bytes_per_ray']}, achieved {bj['roofline']['achieved']} GB/s", ""]
except Exception as e:
    lines += ["", f"(no bench line: {e})", ""]
lines += ["## PMC passes (per dispatch of the traversal kernel over 2^26 rays; mean over dispatches)", "",
          "| counter | mean per dispatch |", "|---|---|"]
vals = {}
for f in sorted(glob.glob(os.path.join(src, "pmc*", "bench_counter_collection.csv"))):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_trace" not in r["Kernel_Name"]:
            continue
        if int(r["Grid_Size"]) < 100000:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # keep only the full-size dispatches (the counting prefix launch is smaller in time but same grid: drop the first)
    for k, v in acc.items():
        v = v[1:] if len(v) > 1 else v
        vals[k] = sum(v) / len(v)
        lines.append(f"| {k} | {vals[k]:.6g} |")
lines.append("")
if "FETCH_SIZE" in vals:
    lines.append(f"FETCH_SIZE is reported in KiB-like units of 1 KB: {vals['FETCH_SIZE']:.4g} -> {vals['FETCH_SIZE']*1024/1e9:.2f} GB per launch as reported.")
if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
    lines.append(f"L2 hit rate = TCC_HIT/(TCC_HIT+TCC_MISS) = {vals['TCC_HIT_sum']/(vals['TCC_HIT_sum']+vals['TCC_MISS_sum']):.3f}")
if "SQ_ACTIVE_INST_VALU" in vals and "SQ_BUSY_CYCLES" in vals:
    pass
os.makedirs("profiles", exist_ok=True)
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
