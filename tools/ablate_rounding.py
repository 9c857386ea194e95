#!/usr/bin/env python3
"""Separates the two error sources of a 16-bit forward at 720p (G14 / G17 inputs): rounding of the WEIGHTS (one fixed
perturbation of the network: its output error is a smooth function of the input, correlated with the signal) and rounding
of the ACTIVATIONS (noise-like).  Baseline = this repo's exact-fp32 path with the full weights (equal to the reference's
output to 1e-7 dB, tests/test_gpu_bf16.py).  Prints mean / rms of the error, its correlation with the PSNR residual
r = out - target and the PSNR delta, per variant."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speinet_amd.speinet import SPEINet, default_args                                      # noqa: E402
from speinet_amd.synth import state_dict_template, synth_frames, synth_frames_edges, synth_state_dict   # noqa: E402

dev = torch.device("cuda:0")
sd = synth_state_dict(state_dict_template(), seed=0)



def _to_uint8(t):
    """tensor2numpy of the harness (inference_SPEINet.py:477-482): [0,1] floats -> uint8, rounded."""
    return t.detach().float().mul(255.0).clamp(0, 255).round().to(torch.uint8)


def _psnr_uint8(a, b, shave: int = 4) -> float:
    """PSNR of two uint8 frames with a 4-pixel border removed (inference_SPEINet.py:484-500)."""
    a, b = a[..., shave:-shave, shave:-shave].double(), b[..., shave:-shave, shave:-shave].double()
    mse = ((a - b) ** 2).mean().item()
    return float("inf") if mse == 0 else 20.0 * __import__("math").log10(255.0 / mse ** 0.5)


def rounded(sd, dt):
    return {k: (v.to(dt).to(v.dtype) if v.dtype == torch.float32 and v.dim() >= 2 else v) for k, v in sd.items()}


def make(sd_, prec, corr):
    n = SPEINet(args=default_args())
    n.load_state_dict(sd_, strict=True)
    n = n.to(dev).eval()
    n.precision, n.corr_precision = prec, corr
    return n


def stats(tag, out, base, gt):
    e = (out - base).double() * 255.0
    r = (base - gt).double() * 255.0
    c = (e * r).mean() / (e.pow(2).mean().sqrt() * r.pow(2).mean().sqrt())
    dp = _psnr_uint8(_to_uint8(out), _to_uint8(gt)) - _psnr_uint8(_to_uint8(base), _to_uint8(gt))
    pred = 4.3429 * (2 * (e * r).mean() + e.pow(2).mean()) / r.pow(2).mean()
    print(f"  {tag:44s} mean e {e.mean():+.4f}  rms e {e.pow(2).mean().sqrt():.4f} levels  corr(e,r) {c:+.4f}  dPSNR {dp:+.2e} dB "
          f"(-{pred:.2e} predicted from e, r)", flush=True)


for case, seed, kind in (("g14", 1401, "smooth"), ("g17", 1701, "edges")):
    if kind == "edges":
        x, gt = synth_frames_edges(1, 720, 1280, seed=seed)
    else:
        x = synth_frames(1, 720, 1280, seed=seed)
        gt = x[:, 1]
    xd = x.to(dev)
    print(case)
    with torch.no_grad():
        base = make(sd, "f32", "bf16x3")(xd).cpu()
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            stats(f"f32 arithmetic, weights rounded to {name}", make(rounded(sd, dt), "f32", "bf16x3")(xd).cpu(), base, gt)
        wb = make(rounded(sd, torch.bfloat16), "f32", "bf16x3")(xd).cpu()
        ob = make(sd, "bf16", "bf16r")(xd).cpu()
        stats("bf16 mode vs f32 (weights + activations)", ob, base, gt)
        stats("bf16 mode vs f32-with-bf16-weights (activations)", ob, wb, gt)
