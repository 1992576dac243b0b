"""How the two bands of the frame train share the GPU, from a rocprofv3 --kernel-trace run: for the steady-state steps, per
traversal launch its duration, how long it ran alone, beside the other band's traversal launch and beside the other band's
other passes; and the dispatch sequence of a few steps.  usage: band_timeline.py <dir> [skip_fraction] [steps_to_print]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:28], r.get("Queue_Id", "?")))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
nprint = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t0, t1 = rows[0][0], max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t0 + skip * (t1 - t0)]
trace = [r for r in rows if r[2].startswith("k_trace")]
def overlap(a, b): return max(0, min(a[1], b[1]) - max(a[0], b[0]))
tot = alone = with_trace = with_other = 0
for r in trace:
    d = r[1] - r[0]
    wt = sum(overlap(r, o) for o in trace if o is not r)
    wo = sum(overlap(r, o) for o in rows if o is not r and not o[2].startswith("k_trace"))
    tot += d; with_trace += wt; with_other += wo
span = rows[-1][1] - rows[0][0]
print(f"span {span/1e6:.1f} ms, {len(trace)} traversal launches, mean {tot/len(trace)/1e3:.0f} us; share of their time beside another traversal launch "
      f"{with_trace/tot:.2f}, beside other passes {with_other/tot:.2f}; traversal launches cover {tot/span:.2f} of the span (sum of durations)")
# time with 0, 1, 2 traversal launches running
ev = sorted([(s, 1) for s, e, *_ in trace] + [(e, -1) for s, e, *_ in trace])
depth = 0; last = ev[0][0]; hist = {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + t - last; depth += d; last = t
print("share of the span by number of traversal launches running:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
base = rows[0][0]
for r in rows[: nprint * 12]:
    print(f"  q{r[3]:>2s} {r[2]:30s} start {(r[0]-base)/1e3:9.1f} end {(r[1]-base)/1e3:9.1f} dur {(r[1]-r[0])/1e3:8.1f} us")
