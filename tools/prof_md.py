#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` output directory into the tracked summary under profiles/:
   python tools/prof_md.py <rocprof dir> <bench log> <profiles/prefix> "<title>" "<command>"
writes <prefix>_kernel_stats.csv (copy), <prefix>_kernel_stats.md (top kernels) and <prefix>_bench.json (the bench line)."""
import csv
import glob
import json
import shutil
import sys

d, log, prefix, title, cmd = sys.argv[1:6]
stats = glob.glob(f"{d}/*/*_kernel_stats.csv")[0]
shutil.copy(stats, prefix + "_kernel_stats.csv")
line = [l for l in open(log) if l.startswith('{"metric"')][-1]
bench = json.loads(line)
json.dump(bench, open(prefix + "_bench.json", "w"), indent=1)
rows = list(csv.DictReader(open(stats)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
with open(prefix + "_kernel_stats.md", "w") as f:
    f.write(f"# {title}\n\nCommand (on the MI355X box): `{cmd}`\n\n")
    rf = bench["roofline"]
    f.write(f"bench line of the same run: {bench['value']:.2f} {bench['unit']}, {bench['ms_per_step']:.1f} ms/step, "
            f"{rf['kernel']} launch {rf['launch_ms']:.2f} ms by HIP events (profiler attached; the tracked kernel's average below "
            f"must agree).  Total kernel time in the trace {tot / 1e6:.1f} ms.\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:28]:
        name = r["Name"].replace("|", "\\|")[:110]
        f.write(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
print(open(prefix + "_kernel_stats.md").read())
