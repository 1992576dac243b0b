"""Prints the kernel_stats.csv of a rocprofv3 --kernel-trace --stats run.  usage: kstats.py <dir>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:78]:80s} calls {r['Calls']:>6s} total ms {float(r['TotalDurationNs'])/1e6:9.2f} avg ms {float(r['AverageNs'])/1e6:8.3f} {r['Percentage']}%")
