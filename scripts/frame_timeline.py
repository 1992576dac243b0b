"""Prints the dispatch timeline of a window of a rocprofv3 kernel trace (csv): python frame_timeline.py trace.csv [n_launches_back] [n_frames]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_launch')]
s, e = idx[-back], idx[-back + nfr]
t0 = int(rows[s]['Start_Timestamp'])
for r in rows[s:e]:
    n = r['Kernel_Name'].replace('void ', '')[:34]
    st = (int(r['Start_Timestamp']) - t0) / 1e3
    en = (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{n:36s} q{r['Queue_Id']:>2s} start {st:9.1f} end {en:9.1f} dur {en-st:8.1f} us")
print("window", (int(rows[e]['Start_Timestamp']) - t0) / 1e3, "us for", nfr, "frames")
