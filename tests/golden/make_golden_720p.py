#!/usr/bin/env python3
"""G14: the REFERENCE's own forward pass at the bench configuration (1 x 5 x 3 x 720 x 1280, `_forwardbs`, synthetic weights
seed 0) — about five minutes and ~20 GB on the build container's CPU.  Only a subsampled view of the output is committed:
every 8th pixel (3 x 90 x 160 floats) plus the per-channel mean / standard deviation of the full frame; the input is
regenerated from its seed by the test.

Run:  python tests/golden/make_golden_720p.py        (needs /root/reference; writes tests/golden/g14_fwd_720p.npz)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, template_args      # noqa: E402


def main():
    from speinet_amd.synth import synth_frames, synth_state_dict
    ms, *_ = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    net = ms.SPEINet(in_channels=3, n_sequence=3, out_channels=3, n_resblock=3, n_feat=32, device="cpu", args=template_args())
    net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
    net.eval()
    seed = 1401
    x = synth_frames(1, 720, 1280, seed=seed)
    t0 = time.time()
    with torch.no_grad():
        out = net(x)[0]
    print(f"reference forward at 720p: {time.time() - t0:.0f} s, output range [{out.min():.3f}, {out.max():.3f}]")
    np.savez_compressed(os.path.join(HERE, "g14_fwd_720p.npz"), seed=seed, sub=out[:, ::8, ::8].numpy(),
                        mean=out.mean(dim=(1, 2)).numpy(), std=out.std(dim=(1, 2)).numpy())


if __name__ == "__main__":
    main()
