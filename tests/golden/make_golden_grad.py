#!/usr/bin/env python3
"""G19: gradients of the REFERENCE's encoder stack (recons_net.inBlock -> encoder_first -> encoder_second, eval-mode
BatchNorm, model/recons_video_ori.py:26-56, model/block.py:8-140) from its own torch.autograd on the CPU, synthetic
name-keyed weights (seed 0).  Loss = sum(lv3 * r3) + 0.5 sum(lv2 * r2) + 0.25 sum(lv1 * r1) with seeded random r (a linear
functional of all three pyramid levels, so every parameter of the three stages receives a gradient).

Committed: per-parameter gradient L2 norms, every 61st element of each gradient (flattened), the loss value; the test
regenerates inputs, weights and r from the seeds.

Run:  python tests/golden/make_golden_grad.py      (needs /root/reference; writes tests/golden/g19_enc_grad_*.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, template_args      # noqa: E402

STRIDE = 61


def functional_weights(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) for s in shapes]


def main():
    from speinet_amd.synth import synth_frames, synth_state_dict
    ms, *_ = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    net = ms.SPEINet(in_channels=3, n_sequence=3, out_channels=3, n_resblock=3, n_feat=32, device="cpu", args=template_args())
    net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
    net.eval()
    rn = net.recons_net
    for name, seed, h, w in (("g19_enc_grad_40x60", 191, 40, 60), ("g19_enc_grad_100x100", 192, 100, 100)):
        x = synth_frames(1, h, w, seed=seed)[:, 1]
        net.zero_grad()
        lv1 = rn.inBlock(x)
        lv2 = rn.encoder_first(lv1)
        lv3 = rn.encoder_second(lv2)
        r1, r2, r3 = functional_weights(seed + 1000, [lv1.shape, lv2.shape, lv3.shape])
        loss = (lv3 * r3).sum() + 0.5 * (lv2 * r2).sum() + 0.25 * (lv1 * r1).sum()
        loss.backward()
        out = {"seed": seed, "loss": loss.item()}
        n = 0
        for stage in ("inBlock", "encoder_first", "encoder_second"):
            for k, p in getattr(rn, stage).named_parameters():
                key = f"recons_net.{stage}.{k}"
                g = p.grad.reshape(-1)
                out["norm/" + key] = g.norm().item()
                out["sub/" + key] = g[::STRIDE].clone().numpy()
                n += 1
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: loss {loss.item():.6f}, {n} parameter gradients, {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
