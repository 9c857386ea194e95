#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc output directory: python tools/pmc_kernels.py <dir> [substr]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    if sub in k:
        agg[k + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):4d} avg {sum(v) / len(v):16.1f}")
