#!/usr/bin/env python3
"""Per-kernel counter summary of a rocprofv3 --pmc run -> markdown table (tracked under profiles/).

    python tools/pmc_summary.py [--top=N] <rocprof dir> [<rocprof dir> ...] > profiles/rNN_pmc_kernels.md

Columns (per kernel name, averaged over its dispatches): SQ counters as collected; derived where the inputs exist:
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES-like denominator: GRBM_GUI_ACTIVE per XCD x 4 SIMDs is not available per
                kernel, so the ratio is taken against SQ_BUSY_CU_CYCLES when present, else against SQ_WAVE_CYCLES x 4 / waves-in-flight)
  lds_conflict= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (x4 = cycles); SQ_VALU_MFMA_BUSY_CYCLES is cycles
(MI355X_MICROARCH.md, cycle constants).  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled below (gfx950 reports half
of wide coalesced reads).
"""
import collections
import csv
import glob
import re
import sys

import functools
import subprocess


@functools.lru_cache(maxsize=None)
def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    try:    # GNU c++filt predates the _Float16 / __bf16 manglings (DF16_, DF16b): hand it `half` (Dh) and a placeholder, rename after
        out = subprocess.run(["c++filt", name.replace("DF16_", "Dh").replace("DF16b", "Dd")], capture_output=True, text=True, timeout=10).stdout.strip()
        return out.replace("half", "_Float16").replace("decimal64", "__bf16") if out and not out.startswith("_Z") else name
    except (OSError, subprocess.SubprocessError):
        return name


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in [a for a in sys.argv[1:] if not a.startswith('--')]:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(anonymous namespace\)::|void ", "", demangle(r["Kernel_Name"]))
                k = re.sub(r"\(.*", "", k)[:90]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for cs in agg.values() for c in cs})
    rows = []
    for k, cs in agg.items():
        n = max(len(v) for v in cs.values())
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        # sort key: wave cycles per run in an SQ pass; bytes per run in a FETCH_SIZE / WRITE_SIZE pass (those carry no SQ counter:
        # round 2 sorted them all by 0, i.e. reverse-alphabetically, and cut the kernels that carry the traffic off the table)
        key = avg.get("SQ_WAVE_CYCLES", 0) * n or (2 * avg.get("FETCH_SIZE", 0) + avg.get("WRITE_SIZE", 0)) * n
        rows.append((key, k, n, avg))
    rows.sort(key=lambda r: (-r[0], r[1]))
    top = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--top=")), 40))
    print("| kernel | dispatches | " + " | ".join(names) + " | derived |")
    print("|---|---:|" + "---:|" * len(names) + "---|")
    for _, k, n, avg in rows[:top]:
        der = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CU_CYCLES" in avg and avg["SQ_BUSY_CU_CYCLES"]:
            der.append(f"MFMA busy {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * avg['SQ_BUSY_CU_CYCLES']):.1%} of CU-busy SIMD cycles")
        if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
            der.append(f"LDS conflict {avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_LDS_IDX_ACTIVE']:.1%}")
        if "FETCH_SIZE" in avg:
            der.append(f"HBM read {2 * avg['FETCH_SIZE'] / 1024:.1f} MB (x2 corrected), {2 * avg['FETCH_SIZE'] * n / 1024:.0f} MB over the run")
        if "WRITE_SIZE" in avg:
            der.append(f"HBM write {avg['WRITE_SIZE'] / 1024:.1f} MB, {avg['WRITE_SIZE'] * n / 1024:.0f} MB over the run")
        print(f"| `{k}` | {n} | " + " | ".join(f"{avg[c]:.3g}" if c in avg else "" for c in names) + " | " + "; ".join(der) + " |")


if __name__ == "__main__":
    main()
