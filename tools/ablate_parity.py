#!/usr/bin/env python3
"""Where does the throughput mode's error come from?  Runs the full-size golden cases (the REFERENCE's own outputs,
tests/golden/g14..g17) through a list of arithmetic configurations and prints, per configuration and case,
max |err| on the committed 8x8 grid, |dPSNR| against the reference's PSNR (the north-star criterion, 1e-3 dB) and the
number of SearchTransfer arg-max positions that differ from the reference's own arg-max.

    python tools/ablate_parity.py [--cases g14_fwd_720p,...] [--out gpurun_out/ablate_parity.json]

Configurations: the three modes, every stage of the path switched to bf16 on its own over an f32-grade (bf16x3)
remainder, and every storage / fusion knob of the bf16 mode switched off on its own (speinet_amd.ops.Ctx).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speinet_amd.speinet import SPEINet, default_args                                      # noqa: E402
from speinet_amd.synth import state_dict_template, synth_frames, synth_frames_edges, synth_state_dict   # noqa: E402


def _to_uint8(t):
    """tensor2numpy of the harness (inference_SPEINet.py:477-482): [0,1] floats -> uint8, rounded."""
    return t.detach().float().mul(255.0).clamp(0, 255).round().to(torch.uint8)


def _psnr_uint8(a, b, shave: int = 4) -> float:
    """PSNR of two uint8 frames with a 4-pixel border removed (inference_SPEINet.py:484-500)."""
    a, b = a[..., shave:-shave, shave:-shave].double(), b[..., shave:-shave, shave:-shave].double()
    mse = ((a - b) ** 2).mean().item()
    return float("inf") if mse == 0 else 20.0 * __import__("math").log10(255.0 / mse ** 0.5)


GOLDEN = os.path.join(ROOT, "tests", "golden")
BF = {"precision": "bf16"}
HF = {"precision": "f16"}
CONFIGS = [
    ("f32", "f32", "bf16x3", {}),
    ("bf16x3 / corr bf16x3", "bf16x3", "bf16x3", {}),
    ("bf16x3 / corr top2(bf16)", "bf16x3", "top2", {}),
    ("bf16x3 / corr single(bf16)", "bf16x3", "single", {}),
    ("only enc bf16", "bf16x3", "bf16x3", {"stage": {"enc": BF}}),
    ("only swin bf16", "bf16x3", "bf16x3", {"stage": {"swin": BF}}),
    ("only decode bf16", "bf16x3", "bf16x3", {"stage": {"decode": BF}}),
    ("only enc f16", "bf16x3", "bf16x3", {"stage": {"enc": HF}}),
    ("only swin f16", "bf16x3", "bf16x3", {"stage": {"swin": HF}}),
    ("only decode f16", "bf16x3", "bf16x3", {"stage": {"decode": HF}}),
    ("bf16 / corr top2", "bf16", "top2", {}),
    ("bf16 / corr single  (round-1 bench mode)", "bf16", "single", {}),
    ("f16 / corr bf16x3", "f16", "bf16x3", {"stage": {"search": {"precision": "bf16x3"}}}),
    ("f16 / corr top2  (bench mode)", "f16", "top2", {}),
    ("f16 / corr top2, split_decode off (round 2)", "f16", "top2", {"split_decode": False}),
    ("f16, split glue(H/4,H/2) + dec2", "f16", "top2", {"split_decode": False, "stage": {"glue": {"precision": "bf16x3", "commute_any": True}, "dec2": {"precision": "bf16x3", "commute_any": True}}}),
    ("f16, split glue(all) + dec2 resblocks (convT f16)", "f16", "top2", {"stage": {"convt": {"precision": "f16"}}}),
    ("f16, split glue(H/4,H/2) + dec2 resblocks", "f16", "top2", {"split_decode": False, "stage": {"glue": {"precision": "bf16x3", "commute_any": True}, "dec2": {"precision": "bf16x3", "commute_any": True}, "convt": {"precision": "f16"}}}),
    ("f16, split glue(H) + dec2", "f16", "top2", {"split_decode": False, "stage": {"glue1": {"precision": "bf16x3", "commute_any": True}, "dec2": {"precision": "bf16x3", "commute_any": True}}}),
    ("f16 / corr single", "f16", "single", {}),
    ("f16 top2, x1 fp32", "f16", "top2", {"x1_bf16": False}),
    ("f16 top2, fp32 storage", "f16", "top2", {"bf16_storage": False}),
    ("f16 top2, unfused MLP (erf GELU)", "f16", "top2", {"fuse_mlp": False}),
    ("f16 top2, unfused attention", "f16", "top2", {"fuse_attn": False}),
    ("f16 top2, conv after upsample", "f16", "top2", {"commute_upconv": False}),
    ("f16 top2, f16 correlation candidates", "f16", "top2", {"corr_bf16": False}),
    ("f16 top2, ResBlock apply fused into the next conv1 (opt-in)", "f16", "top2", {"fuse_apply": True}),
    ("f16, search stage bf16x3", "f16", "top2", {"stage": {"search": {"precision": "bf16x3", "corr_precision": "bf16x3"}}}),
    ("f16, decode stage bf16x3", "f16", "top2", {"stage": {"decode": {"precision": "bf16x3"}}}),
    ("f16, swin stage bf16x3", "f16", "top2", {"stage": {"swin": {"precision": "bf16x3"}}}),
    ("f16, enc stage bf16x3", "f16", "top2", {"stage": {"enc": {"precision": "bf16x3"}}}),
    ("f16, outBlock + tail bf16x3", "f16", "top2", {"stage": {"out": {"precision": "bf16x3"}}}),
    ("f16, tail conv f32", "f16", "top2", {"stage": {"tail": {"precision": "f32"}}}),
    ("f16, tail conv bf16x3", "f16", "top2", {"stage": {"tail": {"precision": "bf16x3"}}}),
    ("f16, dec2 bf16x3", "f16", "top2", {"stage": {"dec2": {"precision": "bf16x3"}}}),
    ("f16, dec1 bf16x3", "f16", "top2", {"stage": {"dec1": {"precision": "bf16x3"}}}),
    ("f16, out bf16x3", "f16", "top2", {"stage": {"out": {"precision": "bf16x3"}}}),
    ("f16, glue only bf16x3", "f16", "top2", {"stage": {"glue": {"precision": "bf16x3", "commute_any": True}}}),
    ("f16, glue + dec2 bf16x3", "f16", "top2", {"stage": {"glue": {"precision": "bf16x3", "commute_any": True}, "dec2": {"precision": "bf16x3"}}}),
    ("f16, glue + dec2 + tail bf16x3", "f16", "top2", {"stage": {"glue": {"precision": "bf16x3", "commute_any": True}, "dec2": {"precision": "bf16x3"}, "tail": {"precision": "bf16x3"}}}),
    ("f16, dec2+dec1 bf16x3", "f16", "top2", {"stage": {"dec2": {"precision": "bf16x3"}, "dec1": {"precision": "bf16x3"}}}),
    ("f16, tail f32 + x1 fp32", "f16", "top2", {"x1_bf16": False, "stage": {"tail": {"precision": "f32"}}}),
    ("f16, tail f32 + decode bf16x3", "f16", "top2", {"stage": {"tail": {"precision": "f32"}, "decode": {"precision": "bf16x3"}}}),
]


def load_case(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    kind = str(d["kind"]) if "kind" in d else "smooth"
    if "sub" not in d:                        # the small cases (g10_*) keep the reference's whole output
        out = torch.from_numpy(d["out"])
        b, _, h, w = out.shape
        zr = tuple(int(i) for i in d["zero_ref"]) if "zero_ref" in d else ()
        x = synth_frames(b, h, w, seed=int(d["seed"]), zero_ref=zr)
        d["sub"] = d["out"][:, :, ::8, ::8]
        d["psnr"] = np.array([_psnr_uint8(_to_uint8(out[i:i + 1]), _to_uint8(x[i:i + 1, 1])) for i in range(b)])
        return d, x, x[:, 1], zr
    tp = os.path.join(GOLDEN, name + "_target.npz")
    if os.path.exists(tp):                    # synthesised ~32 / ~28 dB ground truth and the reference's PSNR against it
        t = np.load(tp)
        d["target_u8"], d["target_psnr"] = t["target"], t["psnr"]
    b = d["sub"].shape[0]
    h, w = d["sub"].shape[2] * 8, d["sub"].shape[3] * 8
    zr = tuple(int(i) for i in d["zero_ref"])
    if kind == "edges":
        x, gt = synth_frames_edges(b, h, w, seed=int(d["seed"]), zero_ref=zr)
    else:
        x = synth_frames(b, h, w, seed=int(d["seed"]), zero_ref=zr)
        gt = x[:, 1]
    return d, x, gt, zr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="g14_fwd_720p,g17_fwd_720p_edges,g16_fwd_480x640_mixed")
    ap.add_argument("--configs", default="", help="comma-separated substrings; default all")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ablate_parity.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    net = SPEINet(args=default_args())
    net.load_state_dict(synth_state_dict(state_dict_template(), seed=0), strict=True)
    net = net.to(dev).eval()
    want = [c for c in args.configs.split(",") if c]
    rows = []
    for case in args.cases.split(","):
        d, x, gt, zr = load_case(case)
        xd = x.to(dev)
        sub = torch.from_numpy(d["sub"])
        ref_arg = torch.from_numpy(d["arg"]).long() if "arg" in d else None
        margin = torch.from_numpy(d["margin"]) if "margin" in d else None
        ref_s = torch.from_numpy(d["s"]) if "s" in d else None
        for label, prec, corr, knobs in CONFIGS:
            if want and not any(wd in label for wd in want):
                continue
            net.precision, net.corr_precision, net.knobs = prec, corr, knobs
            outs, flips, tight, serr = [], 0, 0, 0.0
            with torch.no_grad():
                si = 0
                for i in range(x.shape[0]):                       # one sample at a time: capture is per SearchTransfer call
                    cap = {}
                    outs.append(net(xd[i:i + 1], capture=cap).cpu())
                    if "s_self" in cap and "s_self" in d:
                        serr = max(serr, (cap["s_self"].cpu() - torch.from_numpy(d["s_self"][0])).abs().max().item())
                    if "arg" in cap and ref_arg is not None:
                        a = cap["arg"].cpu().long()
                        diff = a != ref_arg[si]
                        flips += int(diff.sum())
                        tight += int((diff & (margin[si] >= 1e-5)).sum())     # flips that are NOT reference near-ties
                        serr = max(serr, (cap["s"].cpu() - ref_s[si]).abs().max().item())
                        si += 1
            out = torch.cat(outs)
            err = (out[:, :, ::8, ::8] - sub).abs().max().item()
            dp = max(abs(_psnr_uint8(_to_uint8(out[i:i + 1]), _to_uint8(gt[i:i + 1])) - float(d["psnr"][i])) for i in range(x.shape[0]))
            sdp = [_psnr_uint8(_to_uint8(out[i:i + 1]), _to_uint8(gt[i:i + 1])) - float(d["psnr"][i]) for i in range(x.shape[0])]
            dpt = None
            if "target_u8" in d:
                dpt = [_psnr_uint8(_to_uint8(out[i:i + 1]), torch.from_numpy(d["target_u8"][i:i + 1])) - float(d["target_psnr"][i]) for i in range(x.shape[0])]
            rows.append({"case": case, "config": label, "max_err_grid": err, "dpsnr_db": dp, "dpsnr_signed": sdp, "dpsnr_realistic_signed": dpt,
                         "argmax_flips": flips, "flips_margin_ge_1e-5": tight, "max_s_err": serr})
            print(f"{case:26s} {label:40s} max|err| {err:.2e}  dPSNR " + ",".join(f"{v:+.2e}" for v in sdp) + " dB"
                  + ("  at ~32/28 dB: " + ",".join(f"{v:+.2e}" for v in dpt) if dpt else "") + f"  flips {flips:5d} (non-near-tie {tight})", flush=True)
    net.knobs = {}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
