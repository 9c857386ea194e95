#!/usr/bin/env python3
"""Per-kernel roofline table from four rocprofv3 runs of bench.py (tracked under profiles/):

    python tools/kernel_table.py <kernel-trace --stats dir> <--pmc FETCH_SIZE dir> <--pmc WRITE_SIZE dir> <--pmc SQ dir> [top]
    (frames in the trace = launches of the correlation arg-max kernel, one per frame)

Durations: the kernel trace (one frame in flight, one stream: exclusive per-kernel times).  Bytes: FETCH_SIZE (doubled: gfx950 reports
half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section) and WRITE_SIZE, averaged per dispatch of the same kernel name
in the PMC passes (counter collection serialises kernels).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES).
"""
import collections
import csv
import glob
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import demangle          # noqa: E402  (also prints nothing: guarded below)


def short(name: str) -> str:
    k = re.sub(r"\(anonymous namespace\)::|void ", "", demangle(name))
    return re.sub(r"\(.*", "", k)[:100]


def pmc(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    trace, dfetch, dwrite, dsq = sys.argv[1:5]
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 14
    dur = collections.defaultdict(list)
    f = glob.glob(trace + "/**/*kernel_trace.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    frames = float(sum(len(v) for k, v in dur.items() if k.startswith(("corr_slab_kernel", "corr_argmax_kernel", "corr_diag_kernel<"))) or 1)
    fe, wr, sq = pmc(dfetch), pmc(dwrite), pmc(dsq)
    rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:top]
    tot = sum(sum(v) for v in dur.values())
    print(f"Kernel time per frame {tot / frames / 1e3:.2f} ms over {len(dur)} kernel names ({frames:g} frames in the trace).\n")
    print("| kernel | launches / frame | avg us | ms / frame | HBM read MB | HBM write MB | HBM GB/s | % of 8 TB/s | MFMA busy |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k, v in rows:
        avg = sum(v) / len(v)
        rd = 2.0 * fe.get(k, {}).get("FETCH_SIZE", float("nan")) / 1024.0
        wt = wr.get(k, {}).get("WRITE_SIZE", float("nan")) / 1024.0
        gbs = (rd + wt) * 1e6 / (avg * 1e-6) / 1e9 if avg > 0 else float("nan")
        s = sq.get(k, {})
        busy = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * s["SQ_BUSY_CU_CYCLES"]) if s.get("SQ_BUSY_CU_CYCLES") else float("nan")
        print(f"| `{k}` | {len(v) / frames:.1f} | {avg:.1f} | {sum(v) / frames / 1e3:.2f} | {rd:.1f} | {wt:.1f} | {gbs:.0f} | {gbs / 80:.0f} % | {busy:.1%} |")


if __name__ == "__main__":
    main()
