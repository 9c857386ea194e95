#!/usr/bin/env python3
"""Batched encoder stacks (ops.BMap, gridDim.y = pass): three ResBlocks on k maps per launch at the three 720p levels, hipGraph replay,
one stream; time per pass.  With the tuning build, SPEI_SLAB_CFG_N32 / _N64 / _N128 pick other conv tile shapes (conv_slab16.hip).
    python tools/bench_batching.py [k ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import pack                                             # noqa: E402
from speinet_amd.ops import BMap, Ctx                                    # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict      # noqa: E402

dev = "cuda:0"
sd = synth_state_dict(state_dict_template())
ctx = Ctx("f16", device=dev)
ks = [int(v) for v in sys.argv[1:]] or [1, 7]
for name, prefix, h, w, c in (("lv1", "recons_net.inBlock.", 720, 1280, 32), ("lv2", "recons_net.encoder_first.", 360, 640, 64),
                              ("lv3", "recons_net.encoder_second.", 180, 320, 128)):
    blocks = [pack._to_device(pack.resblock(sd, f"{prefix}{i}."), dev) for i in (1, 2, 3)]
    for k in ks:
        x = BMap(torch.randn(k * h * w, c, device=dev), k, h, w, c)

        def run():
            f = x
            for pk in blocks:
                f = ctx.resblock_batched(f, pk)
            return f
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: 3 ResBlocks, {k} maps per launch: {e0.elapsed_time(e1) / 10 * 1e3:7.0f} us = {e0.elapsed_time(e1) / 10 / k * 1e3:6.0f} us per pass")
