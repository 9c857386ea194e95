#!/usr/bin/env python3
"""Time the ResBlock gate kernels (statistics -> reduce -> maps) and the apply kernel at the three 720p levels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import pack                         # noqa: E402
from speinet_amd.ops import Ctx                      # noqa: E402
from speinet_amd.ops import FMap                                         # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict      # noqa: E402

dev = "cuda:0"
ops = Ctx("bf16", device=dev)
sd = synth_state_dict(state_dict_template())
for name, prefix, h, w, c in (("lv1", "recons_net.inBlock.1.", 720, 1280, 32), ("lv2", "recons_net.encoder_first.1.", 360, 640, 64),
                              ("lv3", "recons_net.encoder_second.1.", 180, 320, 128)):
    pk = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in pack.resblock(sd, prefix).items() if not isinstance(v, pack.GemmW)}
    x1 = FMap(torch.randn(h * w, c, device=dev).bfloat16(), h, w, c)
    for _ in range(3):
        ops.resblock_gates(x1, pk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.resblock_gates(x1, pk)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: gates (stats + reduce + maps) {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
