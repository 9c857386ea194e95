"""The training loss of the reference (`Loss/__init__.py:13-80` with `--loss 1*L1+2*HEM`, option/template.py:11,33;
`Loss/hard_example_mining.py:6-47`): a weighted sum of nn.L1Loss and the hard-example-mining L1.

    Loss("1*L1+2*HEM")(sr, hr) -> scalar tensor (differentiable in sr)

HEM(x, y) = mean |x m - y m| with m = hard | random, both computed without gradient (hard_example_mining.py:14-40):
  hard    per sample, the pixels whose channel-summed |x - y| is GREATER than the value at rank int(0.5 * h * w) of the
          descending sort (ties at the threshold are out);
  random  per sample, int(0.1 * h * w) pixels chosen by `np.random.shuffle` of a 0/1 vector — numpy's GLOBAL generator, as in
          the reference, so `np.random.seed` reproduces the reference's masks draw for draw.
The output is a [B,3,H,W] image: this is parameter-free plumbing on the model's output, evaluated with torch ops on the device
(the sort included); only the shuffle runs on the host, as it does in the reference.  GAN / VGG terms are not built.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class HEM(nn.Module):
    def __init__(self, hard_thre_p: float = 0.5, device="cuda", random_thre_p: float = 0.1):
        super().__init__()
        self.hard_thre_p, self.random_thre_p = hard_thre_p, random_thre_p
        self.device = device

    def hard_mining_mask(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            b, c, h, w = x.shape
            res = (x - y).abs().sum(dim=1, keepdim=True)                                   # [b,1,h,w]
            line = res.view(b, -1)
            thre = line.sort(dim=1, descending=True)[0][:, int(self.hard_thre_p * w * h)]  # [b]
            hard = (res > thre.view(b, 1, 1, 1)).float()
            k = int(self.random_thre_p * w * h)
            rnd = np.zeros((b, h * w), dtype=np.float32)
            for i in range(b):
                rnd[i, :k] = 1.0
                np.random.shuffle(rnd[i])
            rnd = torch.from_numpy(rnd).view(b, 1, h, w).to(x.device)
            return ((hard + rnd) > 0).float()

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        mask = self.hard_mining_mask(x.detach(), y.detach())
        return (x * mask - y * mask).abs().mean()


class Loss(nn.Module):
    """`--loss` strings of L1 / MSE / HEM terms, e.g. the reference's "1*L1+2*HEM"; `log` keeps the per-term sums the reference's
    trainer prints (Loss/__init__.py:63-80)."""

    def __init__(self, spec: str = "1*L1+2*HEM", device="cuda"):
        super().__init__()
        self.terms = []
        for part in spec.split("+"):
            weight, kind = part.split("*")
            if kind == "L1":
                fn = nn.L1Loss()
            elif kind == "MSE":
                fn = nn.MSELoss()
            elif kind == "HEM":
                fn = HEM(device=device)
            else:
                raise NotImplementedError(f"Loss type [{kind}] is not built (L1, MSE, HEM are)")
            self.terms.append((float(weight), kind, fn))
        self.log = []

    def forward(self, sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
        parts = [wgt * fn(sr, hr) for wgt, _, fn in self.terms]
        self.log.append([float(p.detach()) for p in parts])
        return sum(parts)
