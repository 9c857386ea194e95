#!/usr/bin/env python3
"""Time the Swin-block kernels alone at the 720p lv3 size (180x320 tokens): the fused attention branch (round 2's two-window kernel,
one map; round 4's four-window kernel on one map and on the two stacked maps of a frame) and the fused MLP branch.  PREC=bf16|f16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import pack                         # noqa: E402
from speinet_amd.ops import Ctx                      # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict    # noqa: E402

H, W = 180, 320
dev = "cuda:0"
ops = Ctx(os.environ.get("PREC", "f16"), device=dev)
LP = torch.float16 if ops.precision == "f16" else torch.bfloat16
sd = synth_state_dict(state_dict_template())
p = "swin.layers.0.residual_group.blocks.1."
bk = pack._to_device(pack.swin_block(sd, p, 8, 5), dev)
x = torch.randn(H * W, 256, device=dev)
yhat = torch.randn(H * W, 256, device=dev).to(LP)
out = torch.empty_like(x)


def timeit(name, fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / n * 1e3:7.1f} us")


a2 = ops.replace(attn_win4=False)
xx, yy = torch.randn(2 * H * W, 256, device=dev), torch.randn(2 * H * W, 256, device=dev).to(LP)
oo = torch.empty_like(xx)
for shift in (0, 2):
    timeit(f"attn_fused (2 windows) shift={shift}", lambda: a2.attn_fused(x, yhat, bk, H, W, shift, out))
    timeit(f"attn_win4 1 map        shift={shift}", lambda: ops.attn_fused(x, yhat, bk, H, W, shift, out))
    timeit(f"attn_win4 2 maps       shift={shift}", lambda: ops.attn_fused(xx, yy, bk, H, W, shift, oo))
timeit("mlp_fused", lambda: ops.mlp_fused(x, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out))
x2 = torch.randn(2 * H * W, 256, device=dev)
out2 = torch.empty_like(x2)
timeit("mlp_fused 2x tokens", lambda: ops.mlp_fused(x2, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out2))
for shift in (0, 2):
    timeit(f"attn + mlp shift={shift}", lambda: a2.mlp_fused(a2.attn_fused(x, yhat, bk, H, W, shift, out), bk["w1"], bk["b1"], bk["w2"], bk["b2"], out))
# the 256 -> 256 channel 3x3 convolution of the Swin body (RSTB tail) on the frame's two stacked maps: persistent kernel / slab kernel
from speinet_amd.ops import BMap                     # noqa: E402
cw = pack.PackedW(torch.randn(9, 256, 256) * 0.03, dev)
cb = torch.randn(256, device=dev)
cx, cr = BMap(xx, 2, H, W, 256), BMap(torch.randn_like(xx), 2, H, W, 256)
timeit("conv3x3 256 pipe, 2 maps", lambda: ops.igemm_batched(cx, cw, cb, 256, 3, residual=cr, out=cr))
slab3 = ops.replace(conv3_pipe=False)
timeit("conv3x3 256 slab, 2 maps", lambda: slab3.igemm_batched(cx, cw, cb, 256, 3, residual=cr, out=cr))
