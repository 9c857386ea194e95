"""LD sharpness detector at inference time: when a clip has no `label/<clip>.npy`, the harness
computes six focus measures per frame on the GPU and labels a frame sharp iff a logistic regression says so
(reference inference_SPEINet.py:177-189, 315-322, 349-353).

The arithmetic runs in speinet_amd/csrc/detector.hip; this module only allocates buffers and applies the 7-number
logistic regression.  The weights are those of LD_detector/pickle/LogisticRegression_0.5_11.pkl, read from the raw
pickle bytes with `pickletools` (the pickle is never loaded) and cross-checked against LD_detector/output.csv:158.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

FEATURES = ("LAP1", "MIS3", "WAV1", "GRA7", "STA3", "DCT3")
LR_COEF = (-0.11971818612047416, -1.2293425023576632, 0.0044214112366378735, -0.042858891031731176,
           0.12867998448379486, 1.5577974574202265)
LR_INTERCEPT = -1.5940041517368388


def focus_measures(frames: torch.Tensor, kernel_size: int = 11) -> torch.Tensor:
    """frames [N,3,H,W] float32 with 0..255 values on a ROCm device -> [N,6] float32 (order of FEATURES)."""
    if not frames.is_cuda:
        raise RuntimeError("speinet_amd.detector runs on MI355X only (HIP kernels); there is no CPU path")
    frames = frames.contiguous().float()
    n, c, h, w = frames.shape
    assert c == 3
    lib = _lib.lib()
    with torch.cuda.device(frames.device):       # kernels launch on the current device: make it the tensor's
        st = C.c_void_p(torch.cuda.current_stream(frames.device).cuda_stream)
        gray = torch.empty(n, h, w, device=frames.device)
        _lib.check(lib.spei_det_gray(C.c_void_p(frames.data_ptr()), C.c_void_p(gray.data_ptr()), n, h, w, st), "spei_det_gray")
        return gray_focus_measures(gray, kernel_size)


def gray_focus_measures(gray: torch.Tensor, kernel_size: int = 11) -> torch.Tensor:
    if not gray.is_cuda:
        raise RuntimeError("speinet_amd.detector runs on MI355X only (HIP kernels); there is no CPU path")
    gray = gray.contiguous().float()
    n, h, w = gray.shape
    lib = _lib.lib()
    with torch.cuda.device(gray.device):
        st = C.c_void_p(torch.cuda.current_stream(gray.device).cuda_stream)
        out = torch.empty(n, 6, device=gray.device)
        ws = torch.empty(lib.spei_det_ws_floats(n, h, w, kernel_size), device=gray.device)
        _lib.check(lib.spei_det_features(C.c_void_p(gray.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), n, h, w,
                                         kernel_size, st), "spei_det_features")
    return out


def predict(features) -> np.ndarray:
    """sklearn LogisticRegression.predict on the six measures: 1 (sharp) iff w.f + b > 0."""
    f = features.detach().double().cpu().numpy() if torch.is_tensor(features) else np.asarray(features, dtype=np.float64)
    return ((f @ np.asarray(LR_COEF) + LR_INTERCEPT) > 0).astype(np.int64)
