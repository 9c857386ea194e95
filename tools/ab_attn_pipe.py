#!/usr/bin/env python3
"""Bit-compare the persistent attention kernel against the one-tile-per-workgroup kernel (tuning build: SPEI_ATTN_PIPE=0 selects the
latter).  Run twice:  SPEI_ATTN_PIPE=0 python tools/ab_attn_pipe.py save ;  python tools/ab_attn_pipe.py check"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import pack                         # noqa: E402
from speinet_amd.ops import Ctx                      # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict    # noqa: E402

dev = "cuda:0"
mode = sys.argv[1]
path = "gpurun_out/ab_attn_pipe.pt"
sd = synth_state_dict(state_dict_template())
bk = pack._to_device(pack.swin_block(sd, "swin.layers.0.residual_group.blocks.1.", 8, 5), dev)
outs = {}
for prec in ("f16", "bf16"):
    ops = Ctx(prec, device=dev)
    lp = torch.float16 if prec == "f16" else torch.bfloat16
    for (B, H, W) in ((1, 5, 5), (1, 10, 15), (3, 15, 25), (2, 35, 20), (2, 180, 320), (1, 360, 640)):
        for shift in (0, 2):
            g = torch.Generator(device="cpu").manual_seed(100 * H + W + shift)
            x = (torch.randn(B * H * W, 256, generator=g) * 1.5 + 0.3).to(dev)
            y = torch.randn(B * H * W, 256, generator=g).to(dev).to(lp)
            o = ops.attn_fused(x, y, bk, H, W, shift, torch.empty_like(x))
            xi = x.clone()
            oi = ops.attn_fused(xi, y, bk, H, W, shift, xi)          # in place
            assert torch.equal(o, oi), ("in place differs", prec, B, H, W, shift)
            assert torch.isfinite(o).all()
            outs[f"{prec}_{B}_{H}_{W}_{shift}"] = o.cpu()
if mode == "save":
    torch.save(outs, path)
    print("saved", len(outs))
else:
    ref = torch.load(path, weights_only=True)
    bad = 0
    for kx, v in outs.items():
        d = (v - ref[kx]).abs().max().item()
        eq = torch.equal(v, ref[kx])
        if not eq:
            bad += 1
        print(f"{kx:24s} {'bit-identical' if eq else 'DIFFERS'}  max |diff| {d:.3e}  (|out| max {v.abs().max():.2f})")
    print("all bit-identical" if bad == 0 else f"{bad} cases differ")
