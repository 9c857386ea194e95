#!/usr/bin/env python3
"""Where a window's time goes in the harness: summary of a rocprofv3 kernel (+ memory-copy) trace of tools/harness_bench.py.

    python tools/harness_timeline.py <rocprof dir> [windows to skip at the start of the timed pass]

The trace holds two passes over the clip (one untimed, one timed: inference.harness_throughput); the LAST `n` correlation launches
(n = half of them) are the timed pass.  Printed: the span per window, the time with 0 / 1 / 2 / 3+ kernels resident, kernel time per
window by family and by stream, the copies (count, bytes, time) per window, the longest idle gaps and what precedes them.
"""
import collections
import csv
import glob
import re
import sys

import gzip
import os
import subprocess

d = sys.argv[1]
if d.endswith(".gz"):                       # the compact copy written below by an earlier run on the GPU box
    rows = list(csv.DictReader(gzip.open(d, "rt")))
else:
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = sorted({r["Kernel_Name"] for r in rows})
    try:                                    # trace names are mangled
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        table = dict(zip(names, dem)) if len(dem) >= len(names) else {}
    except OSError:
        table = {}
    for r in rows:
        r["Kernel_Name"] = re.sub(r"\(anonymous namespace\)::|^void ", "", table.get(r["Kernel_Name"], r["Kernel_Name"]))
    keep = os.environ.get("TRACE_COMPACT")
    if keep:                                # small enough for gpurun_out: re-analyse offline
        with gzip.open(keep, "wt") as o:
            w = csv.writer(o)
            w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id"])
            for r in rows:
                w.writerow([re.sub(r"\(.*", "", r["Kernel_Name"])[:90], r["Start_Timestamp"], r["End_Timestamp"], r.get("Stream_Id", "?")])
corr = [i for i, r in enumerate(rows) if re.search(r"corr_(slab|diag)_kernel", r["Kernel_Name"])]
if len(corr) < 12:
    print(f"only {len(corr)} correlation launches in the trace; kernel names seen: {sorted({r['Kernel_Name'][:60] for r in rows})[:40]}")
    sys.exit(1)
n = len(corr) // 2
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a, b = corr[len(corr) - n + skip], corr[-2]
nwin = (len(corr) - 2) - (len(corr) - n + skip)
seg = rows[a:b]
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
print(f"{nwin} windows, {(t1 - t0) / 1e6 / nwin:.2f} ms per window, {len(seg) / nwin:.0f} kernels per window")


def fam(name):
    for k in ("corr_diag_kernel", "corr_diag_reduce", "corr_diag_final", "corr_slab", "corr_rescore", "attn_fused", "attn_win4", "attn_pipe", "mlp_fused", "mlp_pipe", "conv32_ws", "conv_slab", "igemm_bf16",
              "igemm_f32", "conv5_in", "conv5_out", "gate_", "resblock_apply", "layernorm", "bicubic", "gather_fold", "rl_iter", "patch_invnorm",
              "add_kernel", "split16", "rot90", "any_nonzero"):
        if k in name:
            return k
    if "at::native" in name or "at_cuda" in name or "elementwise" in name:
        return "torch:" + re.sub(r".*at::native::", "", name)[:60]
    return "other:" + name[:60]


ev = []
for r in seg:
    ev.append((int(r["Start_Timestamp"]), 1, r))
    ev.append((int(r["End_Timestamp"]), -1, r))
ev.sort(key=lambda x: (x[0], x[1]))
active, last, hist, gaps, prev_end = 0, t0, collections.Counter(), [], None
for t, dd, r in ev:
    hist[min(active, 3)] += t - last
    if active == 0 and dd == 1 and t - last > 20000:
        gaps.append((t - last, prev_end, r["Kernel_Name"][:70]))
    last = t
    active += dd
    if dd == -1:
        prev_end = r["Kernel_Name"][:70]
print("ms per window with n kernels resident: " + ", ".join(f"{k}: {v / 1e6 / nwin:.2f}" for k, v in sorted(hist.items())))
per_f, per_s, cnt = collections.Counter(), collections.Counter(), collections.Counter()
for r in seg:
    dt = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per_f[fam(r["Kernel_Name"])] += dt
    cnt[fam(r["Kernel_Name"])] += 1
    per_s[r.get("Stream_Id", "?")] += dt
tot = sum(per_f.values())
print(f"kernel time per window {tot / 1e6 / nwin:.2f} ms; by stream: " + ", ".join(f"s{k}: {v / 1e6 / nwin:.2f}" for k, v in per_s.most_common()))
print("| family | launches / window | ms / window |\n|---|---:|---:|")
for k, v in per_f.most_common(40):
    print(f"| `{k}` | {cnt[k] / nwin:.1f} | {v / 1e6 / nwin:.3f} |")
gaps.sort(reverse=True)
print(f"idle gaps > 20 us: {len(gaps)} ({sum(g[0] for g in gaps) / 1e6 / nwin:.2f} ms per window); the longest:")
for g in gaps[:12]:
    print(f"  {g[0] / 1e3:.0f} us after `{g[1]}` before `{g[2]}`")
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if mc:
    cp = [r for r in csv.DictReader(open(mc[0])) if t0 <= int(r["Start_Timestamp"]) < t1]
    by = collections.defaultdict(lambda: [0, 0, 0])
    for r in cp:
        e = by[r.get("Direction", "?")]
        e[0] += 1
        e[1] += int(r.get("Bytes", r.get("Size", 0)) or 0)
        e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, byts, ns) in by.items():
        print(f"copies {k}: {c / nwin:.1f} per window, {byts / 1e6 / nwin:.1f} MB, {ns / 1e6 / nwin:.3f} ms")
