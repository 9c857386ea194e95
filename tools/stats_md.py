#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats summary (…_kernel_stats.csv) -> a markdown table for profiles/ (plus a copy of the csv).

    python tools/stats_md.py <rocprof dir> <out prefix> "<title>" "<command>" ["<note>"] [units-per-run]
"""
import csv
import glob
import shutil
import sys

d, out, title, cmd = sys.argv[1:5]
note = sys.argv[5] if len(sys.argv) > 5 else ""
units = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
shutil.copy(f, out + ".csv")
with open(out + ".md", "w") as o:
    o.write(f"# {title}\n\nCommand (on the MI355X box): `{cmd}`\n\n")
    if note:
        o.write(note + "\n\n")
    o.write(f"Total kernel time {tot / 1e6:.1f} ms over {len(rows)} kernel names" + (f" = {tot / 1e6 / units:.2f} ms per unit ({units:g} units in the trace)" if units else "") + ".\n\n")
    o.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:45]:
        o.write(f"| `{r['Name'][:110]}` | {int(r['Calls'])} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
print(f"{out}.md: {len(rows)} kernels, {tot / 1e6:.1f} ms")
