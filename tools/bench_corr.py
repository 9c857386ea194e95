#!/usr/bin/env python3
"""Time the correlation arg-max kernels alone at the 720p map size (180x320 positions, 128 channels): every mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd.ops import Ctx, FMap                # noqa: E402

H, W = (int(v) for v in os.environ.get("HW", "180x320").split("x"))
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
lr = FMap(torch.randn(H * W, 128, generator=g).to(dev), H, W, 128)
rf = FMap(torch.randn(H * W, 128, generator=g).to(dev), H, W, 128)
fl = 2.0 * 1152 * (H * W) ** 2
for prec, corr in (("f32", "bf16x3"), ("bf16", "bf16x3"), ("bf16", "single"), ("bf16", "top2"), ("f16", "single"), ("f16", "top2")):
    ctx = Ctx(prec, corr, device=dev)
    il, ir = ctx.patch_invnorm(lr), ctx.patch_invnorm(rf)
    plan = ctx.corr_plan(lr, rf, il, ir)
    prof = {"corr_argmax": []}
    for _ in range(3):
        plan.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10 if prec != "f32" else 3
    e0.record()
    for _ in range(n):
        plan.launch(prof)
    e1.record()
    torch.cuda.synchronize()
    k = sum(a.elapsed_time(b) for a, b in prof["corr_argmax"]) / n
    print(f"{prec:5s} / {corr:7s} {plan.kernel:36s} kernel {k:7.3f} ms = {fl / k / 1e9:7.1f} TFLOP/s   whole launch {e0.elapsed_time(e1) / n:7.3f} ms")
