#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) average duration and ms per frame."""
import collections
import csv
import glob
import sys

d, frames = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:48]
    agg[(name, r["Grid_Size_X"], r["Grid_Size_Y"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':50s} {'grid':>14s} {'lds':>6s} {'vgpr':>8s} {'n/frame':>8s} {'avg_us':>9s} {'ms/frame':>9s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print(f"{k[0]:50s} {k[1] + 'x' + k[2]:>14s} {k[3]:>6s} {k[4] + '+' + k[5]:>8s} {len(v) / frames:8.1f} {sum(v) / len(v):9.1f} {sum(v) / frames / 1e3:9.2f}")
print(f"total kernel ms/frame {tot / frames / 1e3:.2f}")
