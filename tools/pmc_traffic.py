#!/usr/bin/env python3
"""HBM traffic of the correlation kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, as
MI355X_MICROARCH.md prescribes) -> profiles/<name>.json.

   python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out json> "<command note>" [key] [commit]
   key: "<precision>/<corr precision>" entry bench.py looks up (default f16/top2)
"""
import csv
import glob
import json
import re
import sys


def counter(d, name, kernel_re):
    f = (glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True))[0]
    vals, kname = [], None
    for r in csv.DictReader(open(f)):
        if re.search(kernel_re, r["Kernel_Name"]) and r["Counter_Name"] == name:
            vals.append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"]
    assert vals, f"no {name} rows for {kernel_re} in {f}"
    return sum(vals) / len(vals), len(vals), kname


def total(d, name):
    f = (glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True))[0]
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name)


CORR = r"corr_(slab|diag)_kernel"        # the candidate kernel itself, not the diagonal form's reduce / final kernels
fetch_kb, n, kname = counter(sys.argv[1], "FETCH_SIZE", CORR)
write_kb, _, _ = counter(sys.argv[2], "WRITE_SIZE", CORR)
# gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts 32-byte units of 64-byte requests
# as one: wide coalesced reads report half their bytes -> double it; WRITE_SIZE is taken as reported.  Units are KiB.
hbm = (2.0 * fetch_kb + write_kb) * 1024.0
KEY = sys.argv[5] if len(sys.argv) > 5 else "f16/top2"
out = {"commit": sys.argv[6] if len(sys.argv) > 6 else "n/a", KEY: {"kernel": kname.replace("(anonymous namespace)::", "")[:80], "launches_averaged": n,
                     "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb, "hbm_bytes_per_launch": hbm,
                     "note": sys.argv[4] + "; FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md "
                             "HBM section); algorithmic compulsory bytes = 2 maps x 14.7 MB (16-bit) + 0.9 MB outputs; the diagonal kernel also writes one "
                             "key pair per (query, diagonal, reference tile): 415 MB at 720p, read back once by its reduce kernel"}}
# per kernel family (bench.py's roofline.families): the same sums restricted to the family's kernels
FAMILIES = {"swin": r"attn_fused_kernel|attn_win4_kernel|attn_pipe_kernel|mlp_fused_kernel|mlp_pipe_kernel",
            "conv": r"conv_slab_kernel|igemm_bf16_kernel|igemm_f32_kernel|conv5_in_kernel|conv5_out_kernel",
            "correlation": r"corr_(slab|diag)_kernel|corr_diag_reduce|corr_diag_final|corr_rescore|corr_argmax",
            "streaming": r"resblock_apply|gate_stats|gate_maps"}


def fam_total(d, name, rx):
    f = (glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True))[0]
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and re.search(rx, r["Kernel_Name"]))


out[KEY]["families_hbm_bytes_per_frame"] = {k: (2.0 * fam_total(sys.argv[1], "FETCH_SIZE", rx) + fam_total(sys.argv[2], "WRITE_SIZE", rx)) * 1024.0 / n
                                            for k, rx in FAMILIES.items()}
# whole frame: every dispatch of the run (n frames = n launches of the correlation kernel; the one-off weight packing of
# the first call is included, < 1 %), same corrections
out[KEY]["path_hbm_bytes_per_frame"] = (2.0 * total(sys.argv[1], "FETCH_SIZE") + total(sys.argv[2], "WRITE_SIZE")) * 1024.0 / n
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
