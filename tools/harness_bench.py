#!/usr/bin/env python3
"""End-to-end harness throughput on a synthetic 720p clip on disk (PNG decode -> selection -> forward -> uint8 -> PSNR/SSIM ->
PNG encode): python tools/harness_bench.py [frames] [precision] [harness flags, e.g. --lanes 1]      (the same measurement bench.py reports as
"harness")"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import inference                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
print(json.dumps(inference.harness_throughput(n, prec, extra_args=sys.argv[3:]), indent=1))
