#!/usr/bin/env python3
"""End-to-end harness throughput on a synthetic 720p clip on disk (PNG decode -> selection -> forward -> uint8 -> PSNR/SSIM ->
PNG encode): python tools/harness_bench.py [frames] [precision]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speinet_amd import inference                     # noqa: E402
from speinet_amd.synth import synth_frames            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
root = tempfile.mkdtemp(prefix="speinet_clip_")
from PIL import Image                                 # noqa: E402
x = synth_frames(1, 720, 1280, seed=5)[0]
for sub in ("blur", "gt"):
    os.makedirs(os.path.join(root, "data", sub, "clip0"), exist_ok=True)
for i in range(n):
    img = (torch.roll(x[i % 5], shifts=(3 * i, -5 * i), dims=(1, 2)).permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
    for sub in ("blur", "gt"):
        Image.fromarray(img).save(os.path.join(root, "data", sub, "clip0", f"{i:06d}.png"), compress_level=1)
os.makedirs(os.path.join(root, "data", "label"), exist_ok=True)
np.save(os.path.join(root, "data", "label", "clip0.npy"), np.asarray([1 if i % 6 == 0 else 0 for i in range(n)]))
for prefetch, tag in ((4, "async I/O"),):
    a = inference.build_args(["--data_path", os.path.join(root, "data"), "--model_path", "synthetic", "--result_path", os.path.join(root, "res"),
                              "--precision", prec, "--prefetch", str(prefetch)])
    inf = inference.Inference(a)
    inf.logger.echo = False
    inf.infer()                                       # warm-up: graph capture, page cache
    torch.cuda.synchronize()
    t0 = time.time()
    tot = inf.infer()
    torch.cuda.synchronize()
    dt = time.time() - t0
    import glob, re
    lines = [l for f in glob.glob(os.path.join(root, "res", "inference_log*")) for l in open(f) if l.startswith(">")][-int(tot[2]):]
    for key in ("pre_time", "forward_time", "post_time"):
        v = [float(re.search(key + r":([\d.e-]+)s", l).group(1)) for l in lines]
        print(f"  mean {key}: {1e3 * sum(v) / len(v):.1f} ms")
    print(f"harness ({tag}, {prec}): {int(tot[2])} frames in {dt:.2f} s = {tot[2].item() / dt:.2f} frames/s incl. PNG decode/encode, PSNR, SSIM")
