#!/usr/bin/env python3
"""Where does a step of the diagonal-sliding correlation kernel spend its time?  Needs the tuning build
(`python -m speinet_amd.build --tuning`): every wave of the eight-wave kernel sums s_memrealtime (100 MHz) intervals per step
section and writes them to the buffer passed through SPEI_STAMP_PTR.  Prints microseconds per step and section (median over waves), 720p map size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H, W = (int(v) for v in os.environ.get("HW", "180x320").split("x"))
dev = "cuda:0"
torch.cuda.set_device(0)
stamps = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
os.environ["SPEI_STAMP_PTR"] = str(stamps.data_ptr())

from speinet_amd.ops import Ctx, FMap                # noqa: E402

g = torch.Generator().manual_seed(3)
lr = FMap(torch.randn(H * W, 128, generator=g).to(dev), H, W, 128)
rf = FMap(torch.randn(H * W, 128, generator=g).to(dev), H, W, 128)
ctx = Ctx("f16", "top2", device=dev)
il, ir = ctx.patch_invnorm(lr), ctx.patch_invnorm(rf)
plan = ctx.corr_plan(lr, rf, il, ir)
for _ in range(3):
    plan.launch()
torch.cuda.synchronize()
stamps.zero_()
prof = {"corr_argmax": []}
plan.launch(prof)
torch.cuda.synchronize()
ms = sum(a.elapsed_time(b) for a, b in prof["corr_argmax"])
s = stamps.view(-1, 8).cpu()
s = s[s[:, 6] > 0]
steps = s[:, 6].float()
names = ["stage loads issued", "fold of the previous row (lagging waves)", "48 MFMAs", "fold of this row (leading waves)",
         "LDS stores of the next rows (incl. the wait for their loads)", "barrier"]
print(f"candidate pass {ms:.3f} ms (stamps included); {s.shape[0]} waves, {steps.median():.0f} steps per wave")
w = torch.arange(s.shape[0]) % 8
for role, sel in (("leading waves (0-3: MFMAs, then the fold of the row)", w < 4), ("lagging waves (4-7: fold of the previous row, then MFMAs)", w >= 4)):
    tot = 0.0
    print(role)
    for k, nme in enumerate(names):
        us = (s[sel, k].float() / steps[sel]) * 0.01
        tot += us.median().item()
        print(f"  {nme:62s} median {us.median():6.3f} us/step   p10 {us.quantile(0.1):6.3f}   p90 {us.quantile(0.9):6.3f}")
    print(f"  sum of medians {tot:.3f} us/step; x steps x 9 rounds of workgroups = {tot * steps.median().item() * 9 / 1000:.3f} ms")
