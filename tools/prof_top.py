#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats run: python tools/prof_top.py <dir> [n] [frames]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
frames = float(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms over {len(rows)} kernels" + (f" = {tot / 1e6 / frames:.2f} ms / frame" if frames else ""))
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['Percentage']):5.1f}%")
