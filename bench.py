#!/usr/bin/env python3
"""Headline benchmark: deblurred 720p frames/s of the SPEINet per-sequence forward pass on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU)

A "step" is one forward of one synthetic [1,5,3,720,1280] window (BASELINE.json configs[1]) per rank, inputs already
resident in HBM.  Frames are independent, so ranks shard by frame with NO data-path collective (weak scaling);
RCCL carries only the barrier, the max-over-ranks time and an all-gather of per-rank output checksums.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     — the dominant kernel (fused correlation arg-max, 37 % of the path's FLOPs, one launch per frame):
                 algorithmic FLOPs per launch / its HIP-event duration measured live in the timed region, against the
                 dense MFMA peak of the dtype in use (MI355X_MICROARCH.md: 2500 TFLOP/s bf16, 157.3 TFLOP/s f32);
                 `path_frac` prices the WHOLE frame (F = 20.64 TFLOP) against the same peak; `traffic` is
                 the kernel's HBM bytes per launch from rocprofv3 PMC passes (profiles/r01_traffic.json, FETCH_SIZE
                 doubled as the guide prescribes for wide coalesced reads on gfx950) or null when that file is absent.
  cpu_baseline — the oracle (CPU restatement of the reference, PyTorch fp32) timed on this host's cores, rank 0, N=1,
                 on a bounded sample (one 360x640 frame), scaled to 720p frames by the FLOP formula of BASELINE.md §2.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 720, 1280
# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters: dense matrix peaks (never the 2:1-sparsity figures)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0, "f16": 2500.0}


def path_flops(h: int, w: int) -> float:
    """F(HW) = 1.410e7*HW + 2304*(HW/16)^2  (`_forwardbs`, reference formulation)."""
    hw = h * w
    return 1.410e7 * hw + 2304.0 * (hw / 16.0) ** 2


def corr_flops(h: int, w: int) -> float:
    n3 = (h // 4) * (w // 4)
    return 2.0 * 1152.0 * n3 * n3     # [N3 x 1152] x [1152 x N3]


def traffic_bytes(precision: str, corr_precision: str, key: str = "hbm_bytes_per_launch"):
    """HBM bytes per launch of the roofline kernel (or per frame of the whole path), measured offline with rocprofv3 --pmc
    (tools/pmc_traffic.py, profiles/r01_traffic.json)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        return d.get(f"{precision}/{corr_precision if precision != 'f32' else 'f32'}", {}).get(key)
    except (OSError, ValueError):
        return None


def cpu_baseline(seed: int) -> dict:
    from oracle import speinet_oracle as O
    from speinet_amd.synth import state_dict_template, synth_frames, synth_state_dict
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    sd = synth_state_dict(state_dict_template(), seed=0)
    sh, sw = 360, 640
    x = synth_frames(1, sh, sw, seed=seed)
    with torch.no_grad():
        t0 = time.time()
        O.forward(x, sd, O.Cfg())
        dt = time.time() - t0
    scale = path_flops(H, W) / path_flops(sh, sw)
    return {"value": 1.0 / (dt * scale), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle (PyTorch fp32 CPU restatement), one {sw}x{sh} _forwardbs frame in {dt:.1f} s, "
                      f"scaled x{scale:.2f} to 720p by F(HW) of BASELINE.md"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16", "f16"], default="f16",
                    help="arithmetic of the GEMM-shaped kernels: f16 (default) = half operands on the 16-bit matrix pipe, the "
                         "configuration that holds the 1e-3 dB PSNR bound; bf16 = BASELINE.json configs[1] to the letter (same "
                         "pipe, same rate, 8-bit significands: 3e-3 dB)")
    ap.add_argument("--corr-precision", choices=["bf16x3", "single", "top2"], default="top2",
                    help="correlation arg-max when --precision is not f32: top2 = 16-bit pass keeping two candidates + exact "
                         "re-score; single = 16-bit winner; bf16x3 = f32-grade scores (bf16 / bf16x3 only), 2.4x the kernel time")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying one hipGraph per frame")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams for the independent neighbour-frame / reference branches of a frame")
    ap.add_argument("--branch", choices=["bs", "b"], default="bs", help="bs: with sharp reference (SearchTransfer); b: SelfTransfer")
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "RANK" in os.environ:            # launched by torch.distributed.run: one rank per GPU over RCCL, also for N = 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world
    else:
        local = 0
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run for N > 1", file=sys.stderr)
    dev = torch.device("cuda", local)

    from speinet_amd.speinet import SPEINet, default_args
    from speinet_amd.synth import state_dict_template, synth_frames, synth_state_dict

    h, w = args.height, args.width
    net = SPEINet(args=default_args())
    net.load_state_dict(synth_state_dict(state_dict_template(), seed=0), strict=True)
    net = net.to(dev).eval()
    net.precision, net.corr_precision = args.precision, args.corr_precision
    net.use_graph = not args.no_graph
    net.streams = args.streams
    # each rank deblurs its own frames (clip shard = rank); two distinct windows alternate so nothing is cached
    frames = [synth_frames(1, h, w, seed=1234 + 17 * rank + i, zero_ref=(0,) if args.branch == "b" else ()).to(dev) for i in range(2)]
    routing = [args.branch == "b"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    checksum = torch.zeros(1, device=dev, dtype=torch.float64)
    with torch.no_grad():
        for i in range(args.warmup):
            net(frames[i % 2], routing=routing)
        prof = {"corr_argmax": []}
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = net(frames[i % 2], routing=routing, profile=None if net.use_graph else prof)
            checksum += out.double().sum()
        barrier()
        dt = time.perf_counter() - t0
    if net.use_graph:
        # HIP events cannot bracket a node inside a replayed graph: time the dominant kernel on eager launches of the
        # same frames on the same stream, right after the timed region
        net.use_graph = False
        with torch.no_grad():
            for i in range(2):
                net(frames[i % 2], routing=routing, profile=prof)
        torch.cuda.synchronize()
    prof = prof["corr_argmax"]
    corr_ms = sum(s.elapsed_time(e) for s, e in prof) / max(1, len(prof))

    from speinet_amd.dist import gather_metrics, max_over_ranks
    tmax = max_over_ranks(dt, dev, dist)
    # the only payload that crosses xGMI: one [checksum, frames] row per rank
    gathered = gather_metrics(torch.cat((checksum, torch.tensor([float(args.steps)], device=dev, dtype=torch.float64))), dist)
    assert torch.isfinite(gathered).all(), "non-finite output"
    assert int(gathered[:, 1].sum().item()) == world * args.steps

    if rank == 0:
        fps = world * args.steps / tmax
        ach = corr_flops(h, w) / (corr_ms * 1e-3) / 1e12 if corr_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        dtype = args.precision if args.precision == "f32" or args.corr_precision == args.precision else f"{args.precision} (correlation {args.corr_precision})"
        path_hbm = traffic_bytes(args.precision, args.corr_precision, "path_hbm_bytes_per_frame") if (h, w) == (H, W) else None
        line = {
            "metric": "deblurred 720p frames/sec", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"SPEINet.forward on synthetic {w}x{h} 5-frame windows, batch 1 per GPU, "
                                   f"{'_forwardbs (SearchTransfer)' if args.branch == 'bs' else '_forwardb (SelfTransfer)'}, "
                                   "synthetic name-keyed weights seed 0", "frames_per_step_per_gpu": 1, "sharding": "frames by rank, no data-path collective"},
            "roofline": {"bound": "mfma", "kernel": "corr_argmax_kernel", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": traffic_bytes(args.precision, args.corr_precision), "launch_ms": corr_ms,
                         "algorithmic_flops_per_launch": corr_flops(h, w),
                         "path_flops_per_frame": path_flops(h, w),
                         "path_frac": path_flops(h, w) * fps / world / 1e12 / peak,
                         "path_hbm_bytes_per_frame": path_hbm,
                         "path_hbm_frac": (path_hbm * fps / world / 8e12) if path_hbm else None},
            "checksum": float(gathered[:, 0].sum().item()),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(1234)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
