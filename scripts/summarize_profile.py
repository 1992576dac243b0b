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
lines += ["## PMC passes (per dispatch of each traversal kernel of a step over 2^26 rays; mean over dispatches)", ""]
# one table per kernel: a step of the certified walk is two or three launches (first pass, second walk of the listed rays by
# wavefront, and the launch that would walk a long list by lane and leaves at once otherwise)
per_kernel = defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "pmc*", "bench_counter_collection.csv"))):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_trace" not in r["Kernel_Name"]:
            continue
        if int(r["Grid_Size"]) < 100000:
            continue
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, k), v in acc.items():
        # the counting build of the prefix launch is a template instance of its own; of the plain instance every dispatch counts
        per_kernel[kn][k] = (sum(v) / len(v), len(v))
vals = defaultdict(float)
for kn in sorted(per_kernel):
    lines += [f"### `{kn}`", "", "| counter | mean per dispatch | dispatches |", "|---|---|---|"]
    for k, (m, cnt) in sorted(per_kernel[kn].items()):
        lines.append(f"| {k} | {m:.6g} | {cnt} |")
        if ", true>" not in kn and "true, false>" not in kn.replace(" ", "")[-14:]:
            vals[k] += m
    lines.append("")
lines += ["Sums over the kernels of one step (counting builds left out):", ""]
if "FETCH_SIZE" in vals:
    lines.append(f"FETCH_SIZE {vals['FETCH_SIZE']:.6g} KB, WRITE_SIZE {vals.get('WRITE_SIZE', 0):.6g} KB per step; bytes past the L2 = 2 x FETCH_SIZE + WRITE_SIZE = "
                 f"{(2 * vals['FETCH_SIZE'] + vals.get('WRITE_SIZE', 0)) * 1024 / 1e9:.2f} GB per step (gfx950 tallies a 128 B fabric read as 64 B).")
if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
    lines.append(f"L2 hit rate = TCC_HIT/(TCC_HIT+TCC_MISS) = {vals['TCC_HIT_sum']/(vals['TCC_HIT_sum']+vals['TCC_MISS_sum']):.3f}")
os.makedirs("profiles", exist_ok=True)
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
