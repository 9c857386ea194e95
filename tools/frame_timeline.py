#!/usr/bin/env python3
"""Timeline of one steady-state frame from a rocprofv3 kernel trace of bench.py: per-stream busy time, time with 0 / 1 / 2 streams
active, the longest single-stream stretches and what runs in them.

    python tools/frame_timeline.py <rocprof dir>      (run on the GPU box right after the trace; prints a summary)
"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
corr = [i for i, r in enumerate(rows) if re.search(r"corr_(slab|diag)_kernel", r["Kernel_Name"])]
a, b = corr[-4], corr[-3]                       # one full frame between two correlation launches, late in the run
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
streams = sorted({r["Stream_Id"] for r in seg})
print(f"frame {(t1 - t0) / 1e6:.2f} ms, {len(seg)} kernels, streams {streams}")
ev = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1, r))
    ev.append((e, -1, r))
ev.sort(key=lambda x: (x[0], x[1]))
active, last, hist = {}, t0, {}
single = []           # (duration, names) of stretches with exactly one kernel resident
cur_start, cur_names = None, []
for t, d, r in ev:
    n = len(active)
    hist[min(n, 3)] = hist.get(min(n, 3), 0) + (t - last)
    last = t
    if d == 1:
        active[id(r)] = r
    else:
        active.pop(id(r), None)
print("time with n kernels resident: " + ", ".join(f"{k}: {v / 1e6:.2f} ms" for k, v in sorted(hist.items())))
per = {}
for r in seg:
    per[r["Stream_Id"]] = per.get(r["Stream_Id"], 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernel time per stream: " + ", ".join(f"{k}: {v / 1e6:.2f} ms" for k, v in per.items()))
# coarse phases: 1 ms buckets, kernels per bucket by family
def fam(n):
    for k in ("corr_slab", "corr_diag", "corr_rescore", "attn_fused", "attn_win4", "attn_pipe", "mlp_fused", "mlp_pipe", "conv32_ws", "conv_slab", "conv5", "gate_", "resblock_apply", "layernorm", "bicubic", "gather_fold", "rl_", "patch_invnorm"):
        if k in n:
            return k
    return "other"
nb = int((t1 - t0) / 1e6) + 1
for bi in range(nb):
    lo, hi = t0 + bi * 1e6, t0 + (bi + 1) * 1e6
    acc = {}
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ov = min(e, hi) - max(s, lo)
        if ov > 0:
            acc[(r["Stream_Id"], fam(r["Kernel_Name"]))] = acc.get((r["Stream_Id"], fam(r["Kernel_Name"])), 0) + ov
    print(f"  ms {bi:2d}: " + "  ".join(f"s{k[0]}:{k[1]} {v / 1e3:.0f}us" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:5]))
