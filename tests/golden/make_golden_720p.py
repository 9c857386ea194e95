#!/usr/bin/env python3
"""G14-G16: the REFERENCE's own forward pass at the full sizes (synthetic weights seed 0): the bench configuration
(1 x 5 x 3 x 720 x 1280, `_forwardbs`), the same with a zeroed reference frame (`_forwardb`), and a mixed-routing batch of
two at 480 x 640 (the BSD frame size) — minutes and ~20 GB on the build container's CPU.  Only a subsampled view of each
output is committed: every 8th pixel, the per-channel mean / standard deviation of the full frame and its PSNR (uint8,
4-pixel crop) against the middle input frame; the inputs are regenerated from their seeds by the tests.

Run:  python tests/golden/make_golden_720p.py [case ...]     (needs /root/reference; writes tests/golden/g1[456]_*.npz)
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
    cases = [("g14_fwd_720p", 1401, 1, 720, 1280, ()), ("g15_fwd_720p_noref", 1501, 1, 720, 1280, (0,)),
             ("g16_fwd_480x640_mixed", 1601, 2, 480, 640, (1,))]
    only = sys.argv[1:]
    for name, seed, b, h, w, zero_ref in cases:
        if only and name not in only:
            continue
        x = synth_frames(b, h, w, seed=seed, zero_ref=zero_ref)
        t0 = time.time()
        with torch.no_grad():
            out = net(x)
        print(f"{name}: reference forward {time.time() - t0:.0f} s, output range [{out.min():.3f}, {out.max():.3f}]")
        # the harness metric of the reference's frame (uint8 round trip, 4-pixel border cropped, inference_SPEINet.py:484-500)
        # against the window's middle input frame as the stand-in target: lets the tests state a PSNR delta at full size
        from oracle import speinet_oracle as O
        psnr = np.array([O.psnr_uint8(O.to_uint8(out[i:i + 1]), O.to_uint8(x[i:i + 1, 1])) for i in range(b)])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, zero_ref=np.array(zero_ref, dtype=np.int64),
                            sub=out[:, :, ::8, ::8].numpy(), mean=out.mean(dim=(2, 3)).numpy(), std=out.std(dim=(2, 3)).numpy(),
                            psnr=psnr)


if __name__ == "__main__":
    main()
