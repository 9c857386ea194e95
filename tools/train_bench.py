#!/usr/bin/env python3
"""Training-step throughput of the swint model on one MI355X (BASELINE.json config 5: trainer/trainer_swint.py on 200x200 crops —
option/template.py:6,22: patch 200, n_sequence 3, batch 20 over the authors' 3 GPUs): forward in train() mode, 1*L1 + 2*HEM,
backward, Adam(lr 1e-4).step(), synthetic crops and name-keyed weights.

    python tools/train_bench.py [--batch 4] [--patch 200] [--steps 5] [--warmup 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...   (N ranks:
        `--batch` crops per rank, gradients averaged over RCCL in 64 MB buckets, speinet_amd.trainer; weak scaling)

Rank 0 prints one JSON line: whole-job crops/s, ms per step (max over ranks) and rank 0's split (forward / loss + backward +
gradient all-reduce / optimizer), measured with HIP events."""
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
from speinet_amd.trainer import allreduce_gradients, broadcast_buffers   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--patch", type=int, default=200)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-baseline", action="store_true",
                    help="also time one training step of the oracle (CPU restatement, fp32, torch autograd) on the host cores: 2 crops")
    ap.add_argument("--model", choices=["swint", "speinet"], default="swint",
                    help="swint: model/swint.py (trainer_swint.py, config 5); speinet: model/speinet.py (trainer_swint_hsa_nsf.py), every "
                         "4th crop without a sharp reference")
    a = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world)
    args = default_args()
    args.n_sequence = 3
    if a.model == "speinet":
        from speinet_amd.speinet import SPEINet as FullNet
        net = FullNet(args=args)
    else:
        net = SPEINet(n_sequence=3, args=args)
    net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
    net = net.to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=0.0)
    loss_fn = Loss("1*L1+2*HEM", device=dev)
    if a.model == "speinet":
        x = synth_frames(a.batch, a.patch, a.patch, seed=7 + 2 * rank, zero_ref=tuple(range(3, a.batch, 4))).contiguous().to(dev)
    else:
        x = synth_frames(a.batch, a.patch, a.patch, seed=7 + 2 * rank)[:, :3].contiguous().to(dev)
    gt = synth_frames(a.batch, a.patch, a.patch, seed=8 + 2 * rank)[:, 1].contiguous().to(dev)
    torch.manual_seed(rank)
    np.random.seed(rank)
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
        allreduce_gradients(net.parameters())
        e[2].record()
        opt.step()
        broadcast_buffers(net)
        e[3].record()
        torch.cuda.synchronize()
        if it >= a.warmup:
            split += [e[i].elapsed_time(e[i + 1]) for i in range(3)]
    split /= a.steps
    ms = float(split.sum())
    if dist is not None:
        t = torch.tensor([ms], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    cpu = None
    if rank == 0 and a.cpu_baseline:
        # the checker as a baseline: the oracle's train-mode graph + torch autograd on the host, fp32, a bounded sample
        import time
        from oracle import speinet_oracle as O
        from speinet_amd.train import drop_path_scales, speinet_drop_path_scales
        threads = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(threads)
        nb = 2
        xs, gs = x[:nb].cpu(), gt[:nb].cpu()
        sd = {k: (v.detach().cpu().clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.detach().cpu())
              for k, v in net.state_dict().items()}
        if a.model == "speinet":
            sc = speinet_drop_path_scales(net.cfg.depths, [bool(v) for v in (xs[:, 3].reshape(nb, -1) == 0).all(dim=1).tolist()], 3)
            calls = sc.get(False, []) + sc.get(True, [])
        else:
            calls = drop_path_scales(net.cfg.depths, nb, 2)
        t0 = time.time()
        with O.train_mode(calls):
            o = (O.forward if a.model == "speinet" else O.forward_swint)(xs, sd, O.Cfg(n_sequence=3))
        Loss("1*L1+2*HEM", device="cpu")(o, gs).backward()
        dt = time.time() - t0
        cpu = {"value": nb / dt, "unit": "crops/s", "cores": threads, "kind": "port",
               "sample": f"oracle train-mode graph + torch autograd (fp32), forward + loss + backward of {nb} crops of {a.patch}x{a.patch} in {dt:.1f} s, no optimizer step"}
    if rank == 0:
        print(json.dumps({"metric": f"training crops/s, {a.model} model, fwd + loss + bwd + Adam", "value": world * a.batch * 1e3 / ms,
                          "unit": "crops/s", "n_gpus": world, "batch_per_gpu": a.batch, "patch": a.patch, "n_sequence": 3, "ms_per_step": ms,
                          "ms": {"forward": float(split[0]), "loss_backward": float(split[1]), "adam": float(split[2])},
                          "loss": float(loss.item()), "dtype": "f32", "data": "synthetic", "scaling": "weak", "cpu_baseline": cpu}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
