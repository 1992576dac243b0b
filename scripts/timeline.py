"""Timeline figures from a rocprofv3 --kernel-trace run: how busy the GPU was and how the launches of the frame pipeline
overlap.  usage: timeline.py <dir> [skip_fraction]   (the first skip_fraction of the span - scene set-up, warm-up - is left out)"""
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= lo]
span = max(r[1] for r in rows) - rows[0][0]
# union of busy intervals and time-weighted concurrency
ev = sorted([(s, 1) for s, e, *_ in rows] + [(e, -1) for s, e, *_ in rows])
busy = 0; depth = 0; last = ev[0][0]; hist = defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    if depth > 0: busy += t - last
    depth += d; last = t
total = sum(e - s for s, e, *_ in rows)
print(f"span {span/1e6:.1f} ms, some kernel running {busy/span:.3f} of it, sum of kernel durations {total/1e6:.1f} ms = {total/span:.2f} kernels at a time")
print("time share by number of kernels running:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
by = defaultdict(list)
for s, e, n, q, st in rows: by[n[:70]].append(e - s)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:72s} calls {len(v):5d} total {sum(v)/1e6:8.1f} ms avg {sum(v)/len(v)/1e3:8.1f} us median {sorted(v)[len(v)//2]/1e3:8.1f} us")
qs = defaultdict(list)
for s, e, n, q, st in rows: qs[q].append((s, e))
for q, v in sorted(qs.items()):
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    b = sum(e - s for s, e in v)
    print(f"  queue {q}: {len(v)} launches, busy {b/span:.3f} of the span, median gap {sorted(gaps)[len(gaps)//2]/1e3 if gaps else 0:.1f} us")
