"""Build the gfx950 shared library in-tree: speinet_amd/libspeinet_hip.so (hipcc cross-compiles without a GPU).

    python -m speinet_amd.build [--force] [--tuning]

Only sources that changed since their object file are recompiled (every object depends on csrc/common.h and on
include/speinet_hip.h), up to 8 hipcc processes at a time.  `--tuning` defines SPEI_TUNING: the tile-shape / ablation
knobs of csrc/common.h (`spei_knob`) then read their environment variables; the shipping build has no knobs.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspeinet_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "speinet_hip.h")
STAMP = os.path.join(CSRC, ".build_flags")


def sources() -> list:
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _mtime(p: str) -> float:
    return os.path.getmtime(p) if os.path.exists(p) else -1.0


def build_lib(force: bool = False, verbose: bool = True, tuning: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
    if tuning:
        flags.append("-DSPEI_TUNING")
        flags += os.environ.get("SPEI_EXTRA_FLAGS", "").split()        # compile-time experiments (e.g. -DSPEI_SLAB_EXP=1)
    flag_text = " ".join(flags)
    if _mtime(STAMP) < 0 or open(STAMP).read() != flag_text:
        force = True
    common = max(_mtime(HEADER), max(_mtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h")))
    jobs, objs = [], []
    for src in sources():
        s, o = os.path.join(CSRC, src), os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _mtime(o) < max(_mtime(s), common):
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if jobs or _mtime(LIB) < max(_mtime(o) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        with open(STAMP, "w") as f:
            f.write(flag_text)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
