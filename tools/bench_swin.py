#!/usr/bin/env python3
"""Time the two fused Swin-block kernels (attention branch, MLP branch) alone at the 720p lv3 size (180x320 tokens)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import pack                         # noqa: E402
from speinet_amd.ops import Ctx                      # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict    # noqa: E402

H, W = 180, 320
dev = "cuda:0"
ops = Ctx("bf16", device=dev)
sd = synth_state_dict(state_dict_template())
p = "swin.layers.0.residual_group.blocks.1."
bk = {k: (v.to(dev) if torch.is_tensor(v) else pack.PackedW(v.t, dev)) for k, v in pack.swin_block(sd, p, 8, 5).items()}
x = torch.randn(H * W, 256, device=dev)
yhat = torch.randn(H * W, 256, device=dev).bfloat16()
out = torch.empty_like(x)
for shift in (0, 2):
    for _ in range(3):
        ops.attn_fused(x, yhat, bk, H, W, shift, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attn_fused(x, yhat, bk, H, W, shift, out)
    e1.record()
    torch.cuda.synchronize()
    print(f"attn_fused shift={shift}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
for _ in range(3):
    ops.mlp_fused(x, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.mlp_fused(x, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out)
e1.record()
torch.cuda.synchronize()
print(f"mlp_fused: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
