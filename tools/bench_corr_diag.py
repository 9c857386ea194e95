#!/usr/bin/env python3
"""Time the two candidate kernels of the "top2" correlation at the 720p map size (180x320 positions, 128 channels): the slab
kernel (9 C-long products per score) and the diagonal-sliding kernel (3 C-long row terms shared along the diagonal)."""
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
res = {}
for prec, diag in (("f16", False), ("f16", True), ("bf16", True)):
    ctx = Ctx(prec, "top2", device=dev, corr_diag=diag)
    il, ir = ctx.patch_invnorm(lr), ctx.patch_invnorm(rf)
    plan = ctx.corr_plan(lr, rf, il, ir)
    prof = {"corr_argmax": []}
    for _ in range(3):
        plan.launch()
    torch.cuda.synchronize()
    n = 10
    for _ in range(n):
        plan.launch(prof)
    torch.cuda.synchronize()
    k = sum(a.elapsed_time(b) for a, b in prof["corr_argmax"]) / n
    res[(prec, diag)] = plan.arg.clone()
    print(f"{prec:5s} {plan.kernel:36s} candidate pass {k:7.3f} ms = {fl / k / 1e9:7.1f} TFLOP/s of the bmm's count "
          f"({fl / (3 if diag else 1) / k / 1e9:7.1f} executed)", flush=True)
print("positions whose winner differs, slab vs diag (f16):", (res[("f16", False)] != res[("f16", True)]).sum().item())
