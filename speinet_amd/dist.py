"""Multi-GPU plumbing: frames / clips are independent units, so ranks shard them with NO data-path collective.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
The only traffic is a barrier, a MAX all-reduce of the wall time and ONE all-gather of a small per-rank metrics
vector — tens of bytes per rank.  The reference's single-process DataParallel (weights re-broadcast
every call, `model/__init__.py:19-20`, `inference_SPEINet.py:569`) is deliberately not reproduced.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of independent units (clips or frames) to ranks: unit u goes to rank u % world."""
    return list(range(rank, n_units, world))


def shard_clips_by_length(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-first balancing of clips (kept whole so cross-window reuse stays inside one rank)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += lengths[i]
    return [sorted(v) for v in out]


def gather_metrics(local: torch.Tensor, dist=None) -> torch.Tensor:
    """All-gather one small per-rank vector (e.g. [sum_psnr, sum_ssim, n_frames, checksum]) -> [world, k]."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.unsqueeze(0).clone()
    bufs = [torch.zeros_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, local)
    return torch.stack(bufs)


def max_over_ranks(seconds: float, device, dist=None) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
