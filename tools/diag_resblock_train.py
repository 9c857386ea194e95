#!/usr/bin/env python3
"""Diagnostic: train-mode ResBlock (batch-statistics BatchNorm(1)) on the HIP path vs a float64 torch restatement, per parameter."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import train as T
from speinet_amd.speinet import _ResBlock

def ref_block(x, P, train=True):
    x1 = F.relu(F.conv2d(x, P["main.0.main.0.weight"], P["main.0.main.0.bias"], padding=2))
    x1 = F.conv2d(x1, P["main.1.main.0.weight"], P["main.1.main.0.bias"], padding=2)
    b, c, h, w = x1.shape
    y = x1.mean(dim=(2, 3))
    s = torch.sigmoid(F.linear(F.relu(F.linear(y, P["se.fc.0.weight"], P["se.fc.0.bias"])), P["se.fc.2.weight"], P["se.fc.2.bias"])).view(b, c, 1, 1)
    def bn(t, p):
        return F.batch_norm(t, None, None, P[p + "weight"], P[p + "bias"], True, 0.01, 1e-5)
    xp = x1.permute(0, 3, 2, 1)
    z = torch.cat((xp.max(dim=1)[0].unsqueeze(1), xp.mean(dim=1).unsqueeze(1)), dim=1)
    g1 = bn(F.conv2d(z, P["te.cw.conv.conv.weight"], padding=3), "te.cw.conv.bn.").permute(0, 3, 2, 1)      # [B,1,H,C] -> [B,C,H,1]
    xp = x1.permute(0, 2, 1, 3)
    z = torch.cat((xp.max(dim=1)[0].unsqueeze(1), xp.mean(dim=1).unsqueeze(1)), dim=1)                        # [B,2,C,W]
    g2 = bn(F.conv2d(z, P["te.hc.conv.conv.weight"], padding=2), "te.hc.conv.bn.").permute(0, 2, 1, 3)      # [B,1,C,W] -> [B,C,1,W]
    return x + x1 * (s + g1 + g2)

torch.manual_seed(0)
for B, H, W, C in ((2, 20, 24, 64), (2, 10, 10, 128), (3, 40, 40, 32)):
    blk = _ResBlock(C)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if "bn.weight" in n: p.fill_(0.8)
            if "bn.bias" in n: p.fill_(0.1)
    x = torch.randn(B, C, H, W)
    r = torch.randn(B, C, H, W)
    outs = {}
    for dt in (torch.float64, torch.float32):
        P = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in blk.named_parameters()}
        xr = x.to(dt).clone().requires_grad_(True)
        o = ref_block(xr, P)
        (o * r.to(dt)).sum().backward()
        outs[dt] = (o.detach(), xr.grad, {k: v.grad for k, v in P.items()})
    blk = blk.to("cuda:0").train()
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    xd = rows(x).to("cuda:0").requires_grad_(True)
    o = T.resblock(xd, blk, B, H, W, True)
    (o * rows(r).to("cuda:0")).sum().backward()
    o64, dx64, g64 = outs[torch.float64]
    o32, dx32, g32 = outs[torch.float32]
    rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
    print(f"B{B} {H}x{W} C{C}: out HIP {rel(o.detach(), rows(o64)):.1e} cpu32 {rel(o32, o64):.1e}; dx HIP {rel(xd.grad, rows(dx64)):.1e} cpu32 {rel(dx32, dx64):.1e}")
    for k, p in blk.named_parameters():
        print(f"   {k:32s} HIP {rel(p.grad, g64[k]):.1e}   cpu32 {rel(g32[k], g64[k]):.1e}")
