#!/usr/bin/env python3
"""Where does a fused Swin kernel spend its time?  Needs the tuning build (`python -m speinet_amd.build --tuning`): wave 0 of
every workgroup stamps s_memtime (100 MHz-independent shader clock ticks) at phase boundaries into a buffer whose address is
passed through SPEI_STAMP_PTR.  Prints the median per-phase durations in microseconds (at the measured clock) and the spread
of workgroup start / end times across the launch.

    python tools/stamp_phases.py attn|attn4|mlp|mlp2|conv3    (attn: round 2's two-window kernel; attn4: the four-window kernel on two stacked maps)
    python tools/stamp_phases.py conv1|conv2|conv3   (conv_slab: the 5x5 ResBlock conv at 720p level 1 / 2 / 3, f16 in and out;
                                                      stamps: 0 start, 1 slab staged, 2 barrier passed, 3 main loop done, 4 stored)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "attn"
H, W = 180, 320
dev = "cuda:0"
torch.cuda.set_device(0)
nwg = 8192
stamps = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
os.environ["SPEI_STAMP_PTR"] = str(stamps.data_ptr())

from speinet_amd import pack                                            # noqa: E402
from speinet_amd.ops import Ctx                                         # noqa: E402
from speinet_amd.synth import state_dict_template, synth_state_dict     # noqa: E402

ops = Ctx("f16", device=dev)
sd = synth_state_dict(state_dict_template())
p = "swin.layers.0.residual_group.blocks.1."
bk = pack._to_device(pack.swin_block(sd, p, 8, 5), dev)
x = torch.randn(H * W, 256, device=dev)
yhat = torch.randn(H * W, 256, device=dev).half()
out = torch.empty_like(x)
x2, yhat2 = torch.randn(2 * H * W, 256, device=dev), torch.randn(2 * H * W, 256, device=dev).half()
out2 = torch.empty_like(x2)


if which.startswith("ws"):            # the weight-stationary 32-channel kernel on 7 stacked 720p maps: ws32 (fp32 in) / ws16 (16-bit in)
    from speinet_amd.ops import BMap                                   # noqa: E402
    wx = torch.randn(7 * 720 * 1280, 32, device=dev)
    wx = wx.half() if which == "ws16" else wx
    wbm = BMap(wx, 7, 720, 1280, 32)
    wpw = pack.PackedW(torch.randn(25, 32, 32) * 0.03, dev)
    wb = torch.randn(32, device=dev)
if which == "conv3":
    from speinet_amd.ops import BMap                                   # noqa: E402
    c3w = pack.PackedW(torch.randn(9, 256, 256) * 0.03, dev)
    c3b = torch.randn(256, device=dev)
    c3x, c3r = BMap(x2, 2, H, W, 256), BMap(torch.randn_like(x2), 2, H, W, 256)
elif which.startswith("conv"):
    from speinet_amd.ops import FMap                                   # noqa: E402
    ch, hh, ww = {"conv1": (32, 720, 1280), "conv2": (64, 360, 640), "conv3": (128, 180, 320)}[which]
    cx = FMap(torch.randn(hh * ww, ch, device=dev).half(), hh, ww, ch)
    cw = pack.PackedW(torch.randn(25, ch, ch) * 0.05, dev)
    cb = torch.randn(ch, device=dev)
    cout = FMap(torch.empty(hh * ww, ch, device=dev, dtype=torch.float16), hh, ww, ch)


def run():
    if which.startswith("ws"):
        ops.igemm_batched(wbm, wpw, wb, 32, 5, act=1, out_dtype=torch.float16)
    elif which.startswith("conv") and which != "conv3":
        ops.igemm(cx, cw, cb, ch, ksize=5, out=cout)
    elif which == "attn":
        ops.replace(attn_win4=False).attn_fused(x, yhat, bk, H, W, 2, out)
    elif which == "attn4":
        ops.attn_fused(x2, yhat2, bk, H, W, 2, out2)
    elif which == "conv3":            # the persistent 3x3 / 256-channel kernel on two stacked token maps (stamps: the workgroup's second tile)
        ops.igemm_batched(c3x, c3w, c3b, 256, 3, residual=c3r, out=c3r)
    elif which == "mlp2":
        ops.mlp_fused(x2, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out2)
    else:
        ops.mlp_fused(x, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out)


for _ in range(5):
    run()
torch.cuda.synchronize()
stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
s = stamps.cpu()
used = s[:, 0] > 0
s = s[used]
n = s.shape[0]
t0 = s[:, 0].min()
span = (s.max() - t0).item()
tick_us = 0.01               # s_memrealtime: 100 MHz reference clock, common to all XCDs
print(f"{which}: {us:.1f} us by HIP events, {n} workgroups, first-start..last-end {span * tick_us:.1f} us")
if (which.startswith("mlp") or which == "conv3") and (s[:, 8] > 0).any():         # the persistent MLP kernel also stamps the shader clock (slots 8..15)
    clk = s[:, 8:].clone()
    s = s[:, :8]
    for i in range(1, 8):
        if (s[:, i] > 0).any() and (clk[:, i] > 0).any():
            j = max(k for k in range(i) if (s[:, k] > 0).any())
            f = (clk[:, i] - clk[:, j]).float() / ((s[:, i] - s[:, j]).float() * tick_us)
            print(f"  shader clock over phase {j}->{i}: median {f.median():.0f} MHz")
last = max(i for i in range(s.shape[1]) if (s[:, i] > 0).any())
for i in range(1, last + 1):
    if not (s[:, i] > 0).any():
        continue
    j = max(k for k in range(i) if (s[:, k] > 0).any())
    d = (s[:, i] - s[:, j]).float() * tick_us
    print(f"  phase {j}->{i}: median {d.median():7.2f} us   p10 {d.quantile(0.1):7.2f}   p90 {d.quantile(0.9):7.2f}")
print("  offsets from stamp 0 (median): " + ", ".join(f"{i}: {((s[:, i] - s[:, 0]).float() * tick_us).median():.2f}" for i in range(1, last + 1) if (s[:, i] > 0).any()))
tot = (s[:, last] - s[:, 0]).float() * tick_us
st = (s[:, 0] - t0).float() * tick_us
print(f"  workgroup lifetime: median {tot.median():.2f} us (p10 {tot.quantile(0.1):.2f}, p90 {tot.quantile(0.9):.2f}); start times: "
      f"p25 {st.quantile(0.25):.1f} p50 {st.quantile(0.5):.1f} p75 {st.quantile(0.75):.1f} max {st.max():.1f} us")
