"""Prints the dispatch timeline of one rendered frame from a rocprofv3 kernel trace (csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_launch')]
s, e = idx[-3], idx[-2]
t0 = int(rows[s]['Start_Timestamp'])
for r in rows[s:e]:
    n = r['Kernel_Name'][:44]
    print(f"{n:46s} q{r['Queue_Id']} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
print("frame period", (int(rows[e]['Start_Timestamp']) - t0) / 1e3, "us")
