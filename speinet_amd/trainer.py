"""The training step around the model (reference trainer/trainer_swint.py:16-57 `Trainer_SWINT.train`, trainer/trainer.py:7-44):
Adam over all parameters, `out = model(input); zero_grad; loss(out, gt).backward(); step`, one process per GPU.

Multi-GPU: the reference wraps the model in nn.DataParallel (model/__init__.py:19-20): one process, the batch scattered over the
GPUs, outputs gathered and the loss taken on device 0, so every parameter receives the gradient of the mean over the WHOLE batch
and only replica 0's BatchNorm buffers persist.  Here each rank is a process with its own equally sized share of the batch
(`torch.distributed`, backend "nccl" = RCCL over xGMI): the local loss is the mean over the local share, the gradients are
averaged over the ranks in a few large buckets (one all-reduce per ~64 MB: xGMI rings are per-link bound, few large messages beat
many small ones; 123 MB of fp32 gradients in total), and rank 0's BatchNorm(1) buffers are broadcast after the step — the same
update as the reference's, with no parameter server and no per-layer collectives.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20) -> int:
    """Average `.grad` of every parameter over the ranks of `group`, in buckets of about `bucket_bytes`.  A parameter without a
    gradient on this rank contributes zeros (every rank must issue the same collectives); a parameter without a gradient on ANY
    rank keeps `.grad = None` — the reference's single-process step leaves it None and Adam skips it (no weight decay, no stale
    momentum: SearchTransfer.search1/2 are never used in forward, the SelfTransfer convs only on steps with a reference-less
    sample).  One extra all-reduce (MAX) of a has-gradient byte per parameter decides that.  Returns the number of all-reduces."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    plist = [p for p in params if p.requires_grad]
    if not plist:
        return 0
    has = torch.tensor([0 if p.grad is None else 1 for p in plist], dtype=torch.int32, device=plist[0].device)
    dist.all_reduce(has, op=dist.ReduceOp.MAX, group=group)
    any_grad = dict(zip((id(p) for p in plist), has.tolist()))
    buckets: List[List[torch.nn.Parameter]] = [[]]
    size = 0
    for p in plist:
        nbytes = p.numel() * p.element_size()
        if buckets[-1] and size + nbytes > bucket_bytes:
            buckets.append([])
            size = 0
        buckets[-1].append(p)
        size += nbytes
    n = 0
    for b in buckets:
        if not b:
            continue
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in b:
            g = flat[off:off + p.numel()].view_as(p)
            if not any_grad[id(p)]:
                pass                                          # no rank produced a gradient: stays None
            elif p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()
        n += 1
    return n + 1


def seed_rank(seed: int, group=None) -> int:
    """Seed the generators the training step draws from — torch's CPU generator (DropPath factors, train.drop_path_scales) and numpy's
    global generator (the HEM random mask, as in the reference's Loss/hard_example_mining.py) — with `seed + rank`, so that the ranks
    draw DIFFERENT masks for their shares of the batch, as the samples of one nn.DataParallel batch do in the reference.  Returns the
    seed used."""
    import numpy as np
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    torch.manual_seed(seed + rank)
    np.random.seed((seed + rank) % (1 << 32))
    return seed + rank


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Rank `src`'s buffers (the BatchNorm(1) running statistics of the gates) to every rank: nn.DataParallel keeps replica 0's."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    bufs = [b for b in module.buffers() if b.is_floating_point() and b.numel() > 0]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b).to(b.dtype))
        off += b.numel()


class Trainer:
    """One optimizer step at a time; `step(input, gt)` returns the (local) loss value.

    model       speinet_amd.swint.SPEINet (train() is called here) — or any nn.Module whose forward is differentiable
    loss        speinet_amd.loss.Loss (or any callable (sr, hr) -> scalar tensor)
    lr, weight_decay   optim.Adam arguments (option/__init__.py:62-75; template: lr 1e-4)
    """

    def __init__(self, model: torch.nn.Module, loss, lr: float = 1e-4, weight_decay: float = 0.0, group=None,
                 optimizer: Optional[torch.optim.Optimizer] = None):
        self.model, self.loss, self.group = model, loss, group
        self.optimizer = optimizer or torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)

    def step(self, input: torch.Tensor, gt: torch.Tensor) -> float:
        self.model.train()
        out = self.model(input)
        self.optimizer.zero_grad()
        loss = self.loss(out, gt)
        loss.backward()
        allreduce_gradients(self.model.parameters(), self.group)
        self.optimizer.step()
        broadcast_buffers(self.model, 0, self.group)
        return float(loss.detach())
