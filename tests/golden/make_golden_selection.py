#!/usr/bin/env python3
"""G11/G12 — golden vectors for the host-side selection logic and tensor <-> uint8 conversions, produced by calling
the REFERENCE's own `Inference` methods unbound (inference_SPEINet.py:239-313, 431-482).

Run: python tests/golden/make_golden_selection.py   (needs /root/reference).  Output: tests/golden/g11_selection.json,
g12_convert.npz.  Modules the script imports at top level but that these methods never touch (cv2, imageio,
torchvision, ptwt, pywt) are absent here and replaced by empty stand-ins; nothing of the reference's
logic is replaced.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, REF  # noqa: E402


def main():
    import_reference()
    for n in ("imageio", "torchvision", "pywt"):
        sys.modules.setdefault(n, types.ModuleType(n))
    pt = types.ModuleType("ptwt")
    pt.wavedec2 = None
    sys.modules.setdefault("ptwt", pt)
    sys.path.insert(0, REF)
    import inference_SPEINet as inf

    self = types.SimpleNamespace(border=True, n_seq=3)
    self.return_BlurryIndices = lambda d, dist=7: inf.Inference.return_BlurryIndices(self, d, dist)
    patterns = {
        "none_sharp": [0] * 12,
        "one_sharp": [0, 0, 0, 1, 0, 0, 0, 0, 0, 0],
        "alternating": [1, 0] * 8,
        "all_sharp": [1] * 9,
        "long_gaps": [1] + [0] * 11 + [1] + [0] * 15 + [1, 0, 0],
        "short_clip": [0, 1, 0, 0, 1],
        "sharp_ends": [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1],
        "dense": [0, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 0, 0, 1, 0],
        "random40": list(np.random.RandomState(5).binomial(1, 0.3, size=40).astype(int)),
        "random25_sparse": list(np.random.RandomState(6).binomial(1, 0.08, size=25).astype(int)),
    }
    out = {}
    for name, lab in patterns.items():
        lab = [int(v) for v in lab]
        pre, sub = inf.Inference.return_BlurryIndices(self, list(lab))
        frames = [f"clip/{i:06d}.png" for i in range(len(lab))]
        out[name] = {"labels": lab, "pre": [int(v) for v in pre], "sub": [int(v) for v in sub]}
        for border in (True, False):
            self.border = border
            seqs, padded = inf.Inference.gene_seq(self, list(frames), 3)
            pre_w, sub_w = inf.Inference.gene_seq_nsf(self, np.array(lab), 3)
            out[name][f"border_{border}"] = {"seqs": seqs, "padded": padded,
                                             "pre_windows": [[int(v) for v in w] for w in pre_w],
                                             "sub_windows": [[int(v) for v in w] for w in sub_w]}
        self.border = True
    json.dump(out, open(os.path.join(HERE, "g11_selection.json"), "w"), indent=0)
    print("g11_selection.json:", len(out), "patterns")

    # G12: numpy2tensor / tensor2numpy / calc_PSNR
    import torch
    r = np.random.RandomState(12)
    imgs = [r.randint(0, 256, size=(20, 24, 3)).astype(np.uint8) for _ in range(5)]
    t = inf.Inference.numpy2tensor(self, imgs)
    o = torch.from_numpy(r.randn(1, 3, 20, 24).astype(np.float32) * 0.4 + 0.5)
    u8 = inf.Inference.tensor2numpy(self, o)
    gt = r.randint(0, 256, size=(20, 24, 3)).astype(np.uint8)
    psnr = inf.Inference.calc_PSNR(self, gt[4:-4, 4:-4], u8[4:-4, 4:-4])
    np.savez_compressed(os.path.join(HERE, "g12_convert.npz"), imgs=np.stack(imgs), tensor=t.numpy(), o=o.numpy(), u8=u8, gt=gt,
                        psnr=np.float64(psnr))
    print("g12_convert.npz: psnr", psnr)


if __name__ == "__main__":
    main()
