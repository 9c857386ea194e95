#!/usr/bin/env python3
"""The two cross-window-attention SwinIR calls of a frame (model/speinet.py:85: swin(f_mid, feat_i) for the two neighbour frames) at
720p: one after the other on one stream, and side by side on two (the frame path's layout); hipGraph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import engine, pack                                     # noqa: E402
from speinet_amd.ops import Ctx, FMap                                    # noqa: E402
from speinet_amd.speinet import default_args                              # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict      # noqa: E402

dev = "cuda:0"
H, W = 180, 320
sd = synth_state_dict(state_dict_template())
from types import SimpleNamespace
a = default_args()
cfg = SimpleNamespace(n_sequence=3, n_feat=32, n_resblock=3, window_size=5, embed_dim=256, depths=tuple(a.depths), num_heads=tuple(a.num_heads),
                      mlp_ratio=2, rgb_range=1.0, patch_size=200)
P = pack.pack_all(sd, cfg, dev)
ctx = Ctx(os.environ.get("PREC", "f16"), "top2", device=dev, **(eval(os.environ["KNOBS"]) if "KNOBS" in os.environ else {}))
f_mid = FMap(torch.randn(H * W, 128, device=dev), H, W, 128)
feats = [FMap(torch.randn(H * W, 128, device=dev), H, W, 128) for _ in range(2)]
cat = FMap(torch.empty(H * W, 384, device=dev), H, W, 384)
side = torch.cuda.Stream(device=dev)


def pair(two_streams: bool):
    main = torch.cuda.current_stream()
    sx = engine.SwinX(ctx, f_mid, P["swin"])
    if hasattr(engine, "swin_pair") and os.environ.get("PAIR", "1") == "1" and not two_streams:
        engine.swin_pair(ctx, sx, feats, P["swin"], [cat.view(128, 128), cat.view(256, 128)])
        return
    ev = torch.cuda.Event()
    ev.record(main)
    engine.swin(ctx, sx, feats[0], P["swin"], out=cat.view(128, 128))
    if two_streams:
        side.wait_event(ev)
        with torch.cuda.stream(side):
            engine.swin(ctx, sx, feats[1], P["swin"], out=cat.view(256, 128))
        main.wait_stream(side)
    else:
        engine.swin(ctx, sx, feats[1], P["swin"], out=cat.view(256, 128))


for two in (False, True):
    for _ in range(2):
        pair(two)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pair(two)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"two swin calls, {'2 streams' if two else '1 stream '}: {e0.elapsed_time(e1) / 10:.2f} ms")
