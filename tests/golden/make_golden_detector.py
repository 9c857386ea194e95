#!/usr/bin/env python3
"""G13 — golden vectors for the LD-detector focus measures, produced by the REFERENCE's own functions
(inference_SPEINet.py:118-175) on gray frames [4,1,64,80]; WAV1 needs ptwt/pywt (absent) and is not covered."""
import os
import sys
import types

import numpy as np
import torch

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
    r = np.random.RandomState(13)
    yy, xx = np.meshgrid(np.arange(64), np.arange(80), indexing="ij")
    frames = []
    for i in range(4):
        base = 0.5 + 0.3 * np.sin(0.2 * (i + 1) * yy + 0.13 * xx) * np.cos(0.07 * xx * (i + 1))
        frames.append(np.clip(base + (0.02 + 0.05 * i) * r.randn(64, 80), 0, 1))
    g = torch.from_numpy(np.stack(frames)[:, None].astype(np.float32))
    out = {"gray": g.numpy()}
    for k in (11, 7):
        out[f"lap1_k{k}"] = inf.focus_measure_lap1(g, k).numpy()
        out[f"mis3_k{k}"] = inf.focus_measure_mis3(g, k).numpy()
        out[f"gra7_k{k}"] = inf.focus_measure_gra7(g, k).numpy()
        out[f"sta3_k{k}"] = inf.focus_measure_sta3(g, k).numpy()
        out[f"dct3_k{k}"] = inf.focus_measure_dct3(g, k).numpy()
    np.savez_compressed(os.path.join(HERE, "g13_detector.npz"), **out)
    print({k: v.tolist() for k, v in out.items() if k != "gray"})


if __name__ == "__main__":
    main()
