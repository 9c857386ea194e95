#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage (no GPU needed).

    python tools/kernel_resources.py [file.hip ...]      (default: every csrc/*.hip)
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speinet_amd", "csrc")


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def analyse(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-c", src,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return os.path.basename(src), rows


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(8) as ex:
        for f, rows in ex.map(analyse, files):
            print(f"== {f}")
            for r in rows:
                name = re.sub(r"\(anonymous namespace\)::", "", demangle(r["name"]))
                name = re.sub(r"\(.*", "", name)[:70]
                print(f"  {name:70s} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>4} spill {r.get('VGPRs Spill', '?'):>3} "
                      f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} occ {r.get('Occupancy [waves/SIMD]', '?'):>2} LDS {r.get('LDS Size [bytes/block]', '?')}")


if __name__ == "__main__":
    main()
