#!/usr/bin/env python3
"""Time single conv_slab layer shapes (run on the GPU box).  SPEI_SLAB_DBG=1/2/4 ablates staging / MFMA / stores."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speinet_amd import pack
from speinet_amd.ops import Ctx
from speinet_amd.ops import FMap
ops = Ctx(os.environ.get("PREC", "f16"))
LP = torch.float16 if ops.precision == "f16" else torch.bfloat16
dev = "cuda:0"
shapes = [  # name, H, W, Cin, Cout, ks, stride, residual
    ("lv1 conv5 32->32", 720, 1280, 32, 32, 5, 1, False),
    ("lv2 conv5 64->64", 360, 640, 64, 64, 5, 1, False),
    ("lv3 conv5 128->128", 180, 320, 128, 128, 5, 1, False),
    ("swin conv3 256->256", 180, 320, 256, 256, 3, 1, True),
    ("linear 256->512", 57600, 1, 256, 512, 1, 1, False),
    ("linear 256->256 +res", 57600, 1, 256, 256, 1, 1, True),
    ("linear 512->256 +res", 57600, 1, 512, 256, 1, 1, True),
]
IN_BF16 = os.environ.get("IN_BF16", "0") == "1"      # ResBlock conv2: bf16 in, bf16 out
for name, h, w, ci, co, ks, st, res in shapes:
    x = FMap(torch.randn(h * w, ci, device=dev).to(LP if (IN_BF16 and ks == 5) else torch.float32), h, w, ci)
    wt = pack.PackedW(torch.randn(ks * ks, co, ci) * 0.05, dev)
    b = torch.randn(co, device=dev)
    r = FMap(torch.randn(h * w, co, device=dev), h, w, co) if res else None
    out = FMap(torch.empty(h * w, co, device=dev, dtype=LP if ks == 5 else torch.float32), h, w, co)   # as in the ResBlocks
    for _ in range(3):
        ops.igemm(x, wt, b, co, ksize=ks, stride=st, residual=r, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    s.record()
    for _ in range(n):
        ops.igemm(x, wt, b, co, ksize=ks, stride=st, residual=r, out=out)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    fl = 2.0 * h * w * co * ci * ks * ks
    byt = 4.0 * h * w * (ci + co * (2 if res else 1))
    print(f"{name:24s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {byt / us / 1e6:6.2f} TB/s (fp32 in+out)")
