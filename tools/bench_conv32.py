#!/usr/bin/env python3
"""The 32 -> 32 channel 5x5 layer at 720p (the ResBlock convs of inBlock / outBlock): weight-stationary kernel (csrc/conv32_ws16.hip)
against the slab kernel, 1 and 7 stacked maps, fp32 and 16-bit input.  PREC=f16|bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import pack                         # noqa: E402
from speinet_amd.ops import BMap, Ctx                # noqa: E402

H, W = 720, 1280
dev = "cuda:0"
prec = os.environ.get("PREC", "f16")
ws, slab = Ctx(prec, device=dev), Ctx(prec, device=dev, conv32_ws=False)
LP = torch.float16 if prec == "f16" else torch.bfloat16
pw = pack.PackedW(torch.randn(25, 32, 32) * 0.03, dev)
b = torch.randn(32, device=dev) * 0.1


def timeit(name, fn, flops, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:44s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")


for maps in (1, 7):
    for in16 in (False, True):
        x = torch.randn(maps * H * W, 32, device=dev)
        x = x.to(LP) if in16 else x
        bm = BMap(x, maps, H, W, 32)
        fl = 2.0 * 25 * 32 * 32 * H * W * maps
        for name, c in (("weight-stationary", ws), ("slab", slab)):
            timeit(f"{name}, {maps} map(s), {'16-bit' if in16 else 'fp32'} in, 16-bit out", lambda: c.igemm_batched(bm, pw, b, 32, 5, act=1, out_dtype=LP), fl)
