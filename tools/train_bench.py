#!/usr/bin/env python3
"""Training-step throughput of the swint model on one MI355X (BASELINE.json config 5: trainer/trainer_swint.py on 200x200 crops —
option/template.py:6,22: patch 200, n_sequence 3, batch 20 over the authors' 3 GPUs): forward in train() mode, 1*L1 + 2*HEM,
backward, Adam(lr 1e-4).step(), synthetic crops and name-keyed weights.

    python tools/train_bench.py [--batch 4] [--patch 200] [--steps 5] [--warmup 2]

Prints one JSON line: crops/s, ms per step and its split (forward / loss + backward / optimizer), measured with HIP events."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd.loss import Loss                                         # noqa: E402
from speinet_amd.speinet import default_args                             # noqa: E402
from speinet_amd.swint import SPEINet                                    # noqa: E402
from speinet_amd.synth import synth_frames, synth_state_dict             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--patch", type=int, default=200)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = "cuda:0"
    args = default_args()
    args.n_sequence = 3
    net = SPEINet(n_sequence=3, args=args)
    net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
    net = net.to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=0.0)
    loss_fn = Loss("1*L1+2*HEM", device=dev)
    x = synth_frames(a.batch, a.patch, a.patch, seed=7)[:, :3].contiguous().to(dev)
    gt = synth_frames(a.batch, a.patch, a.patch, seed=8)[:, 1].contiguous().to(dev)
    torch.manual_seed(0)
    np.random.seed(0)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    split = np.zeros(3)
    for it in range(a.warmup + a.steps):
        e = [ev() for _ in range(4)]
        e[0].record()
        out = net(x)
        e[1].record()
        opt.zero_grad()
        loss = loss_fn(out, gt)
        loss.backward()
        e[2].record()
        opt.step()
        e[3].record()
        torch.cuda.synchronize()
        if it >= a.warmup:
            split += [e[i].elapsed_time(e[i + 1]) for i in range(3)]
    split /= a.steps
    ms = float(split.sum())
    print(json.dumps({"metric": "training crops/s, swint model, fwd + loss + bwd + Adam", "value": a.batch * 1e3 / ms, "unit": "crops/s",
                      "batch": a.batch, "patch": a.patch, "n_sequence": 3, "ms_per_step": ms,
                      "ms": {"forward": float(split[0]), "loss_backward": float(split[1]), "adam": float(split[2])},
                      "loss": float(loss.item()), "dtype": "f32", "data": "synthetic"}))


if __name__ == "__main__":
    main()
