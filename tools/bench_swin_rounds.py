#!/usr/bin/env python3
"""How do the fused Swin kernels scale with the number of workgroups per CU?  Times attn_fused (2 windows per 256-thread workgroup,
2 workgroups resident per CU) and mlp_fused (128 tokens per 512-thread workgroup, 1 resident) on maps whose workgroup counts are
exact multiples / fractions of the chip's resident slots: if a launch with one workgroup per CU takes as long as one with two, the
residents do not slow each other and the launch time is rounds x workgroup lifetime."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import pack                         # noqa: E402
from speinet_amd.ops import Ctx                      # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict    # noqa: E402

dev = "cuda:0"
ops = Ctx("f16", device=dev)
sd = synth_state_dict(state_dict_template())
bk = pack._to_device(pack.swin_block(sd, "swin.layers.0.residual_group.blocks.1.", 8, 5), dev)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, W in ((40, 80), (80, 80), (80, 160), (120, 160), (160, 160), (160, 240), (160, 320), (180, 320), (200, 320), (240, 320), (320, 320)):
    x = torch.randn(H * W, 256, device=dev)
    yhat = torch.randn(H * W, 256, device=dev).half()
    out = torch.empty_like(x)
    nwin = (H // 5) * (W // 5)
    ta = timeit(lambda: ops.attn_fused(x, yhat, bk, H, W, 2, out))
    tm = timeit(lambda: ops.mlp_fused(x, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out))
    print(f"{H:3d}x{W:3d}: {nwin:5d} windows = {nwin // 2:5d} attention workgroups ({nwin / 2 / 256:5.2f} per CU): {ta:6.1f} us   |   "
          f"{H * W:6d} tokens = {(H * W + 127) // 128:4d} MLP workgroups ({(H * W + 127) // 128 / 256:5.2f} per CU): {tm:6.1f} us", flush=True)
