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


# ---- one command for N GPUs (the reference's multi-GPU entry is one command too, inference_SPEINet.py:234-235) -------
def rank_command(script: str, argv: Sequence[str], nproc: int, port: int, python: str = None, module: bool = False) -> List[str]:
    """The command line that runs `script argv...` as `nproc` ranks of ONE node, one process per GPU:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script argv...`
    (127.0.0.1: the container hostname may not resolve; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* reach the ranks through the env).
    `module=True`: `script` is a module name (`-m speinet_amd.inference`)."""
    import sys
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), *(["-m"] if module else []), script, *argv]


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(script: str, argv: Sequence[str], nproc: int, need_gpus: bool = True, env: dict = None, module: bool = False) -> int:
    """Run `script argv...` as `nproc` ranks and return the worst exit code.  Called by a PARENT that has not touched the GPU (no HIP
    call, no `torch.cuda.is_available()`; counting devices is safe) and never will: the ranks are child processes — nothing is
    re-exec'd — whose stdout / stderr pass through, so rank 0's one JSON line is the parent's output.  Fails loudly, before anything
    starts, when the node has fewer GPUs than ranks."""
    import os
    import subprocess
    import sys
    if nproc < 1:
        raise ValueError(f"launch_ranks: nproc={nproc}")
    if need_gpus:
        have = torch.cuda.device_count()
        if have < nproc:
            print(f"error: {nproc} ranks requested (one per GPU) but this node shows {have} GPU(s)", file=sys.stderr)
            return 2
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL over dmabuf IPC (the host driver supports nothing else)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.call(rank_command(script, argv, nproc, free_port(), module=module), env=e)
