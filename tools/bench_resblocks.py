#!/usr/bin/env python3
"""Time a stack of three ResBlocks at the three 720p levels, with and without the fused apply (Ctx knob fuse_apply)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import engine, pack                                     # noqa: E402
from speinet_amd.ops import Ctx, FMap                                    # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict      # noqa: E402

dev = "cuda:0"
sd = synth_state_dict(state_dict_template())
for name, prefix, h, w, c in (("lv1", "recons_net.inBlock.", 720, 1280, 32), ("lv2", "recons_net.encoder_first.", 360, 640, 64),
                              ("lv3", "recons_net.encoder_second.", 180, 320, 128)):
    blocks = [pack._to_device(pack.resblock(sd, f"{prefix}{i}."), dev) for i in (1, 2, 3)]
    x = FMap(torch.randn(h * w, c, device=dev), h, w, c)
    for fa in (True, False):
        ctx = Ctx("f16", device=dev, fuse_apply=fa)
        for _ in range(3):
            engine._resblocks(ctx, x, blocks)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            engine._resblocks(ctx, x, blocks)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: 3 ResBlocks, fuse_apply={fa}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
