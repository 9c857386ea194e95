"""Host-side sharpness-prior selection: which sharp frames accompany each 3-frame window.

Restates the behaviour of the reference harness (inference_SPEINet.py):
  * `Inference.return_BlurryIndices` :239-313 — per frame, the nearest previous / next frame the detector labelled
    sharp, provided it is closer than `dist` frames; otherwise a deliberately distant index, so that the `> 7` test of
    `infer` (:385-388) zeroes that reference and routes the sample to the no-reference branch;
  * `gene_seq` :431-444 / `gene_seq_nsf` :446-464 — reflect-pad the clip by n_seq//2 and cut sliding windows;
  * window assembly :364-388 — append frame[pre of the FIRST window frame] and frame[sub of the LAST window frame],
    compare frame numbers against the LAST window frame (`frame_numbers[2]`, not the middle one);
  * `numpy2tensor` / `tensor2numpy` / `calc_PSNR` :466-500.
Pure Python / numpy: this is bookkeeping, not arithmetic; it is pinned by tests/golden/g11_selection.json and
g12_convert.npz, which were produced by the reference's own methods.
"""
from __future__ import annotations

import bisect
import math
import os
from typing import Callable, List, Sequence, Tuple

import numpy as np
import torch


def blurry_indices(labels: Sequence[int], dist: int = 7) -> Tuple[List[int], List[int]]:
    """(pre, sub): index of the sharp frame to use before / after each frame of the (already padded) clip."""
    n = len(labels)
    sharp = [i for i, v in enumerate(labels) if v == 1]
    far_hi = lambda i: i + 2 if i < n - 2 else i
    pre, sub = [], []
    if len(sharp) > 1:
        for i in range(n):
            k = bisect.bisect_left(sharp, i)
            if k < len(sharp) and sharp[k] == i:                 # the frame itself is sharp
                pre.append(i)
                sub.append(i)
            elif k == 0:                                         # before the first sharp frame
                f = sharp[0]
                if f - i < dist:
                    pre.append(f)
                    sub.append(f)
                else:
                    pre.append(i - 2 if i > 1 else i)
                    sub.append(far_hi(i))
            elif k == len(sharp):                                # after the last sharp frame
                last = sharp[-1]
                if i - last < dist:
                    pre.append(last)
                    sub.append(last)
                else:
                    pre.append(i - 2)
                    sub.append(far_hi(i))
            else:                                                # between two sharp frames
                a, b = sharp[k - 1], sharp[k]
                pre.append(a if i - a < dist else i - 2)
                sub.append(b if b - i < dist else i + 2)
    else:                                                        # fewer than two sharp frames: neighbours
        for i in range(n):
            pre.append(i - 1 if i > 0 else 0)
            sub.append(i + 1 if i < n - 1 else i)
    # anything that is not a sharp frame becomes an index from the far end of the clip (=> zeroed by the >7 rule)
    sset = set(sharp)
    for lst in (pre, sub):
        for i in range(n):
            if lst[i] not in sset:
                lst[i] = n - 1 if i < n // 2 else 0
    return pre, sub


def reflect_pad(seq: Sequence, n_seq: int) -> list:
    half = n_seq // 2
    seq = list(seq)
    head = seq[1:1 + half][::-1]
    tail = seq[-half - 1:-1][::-1]
    return head + seq + tail


def windows(seq: Sequence, n_seq: int) -> list:
    seq = list(seq)
    return [seq[i:i + n_seq] for i in range(len(seq) - 2 * (n_seq // 2))]


def gene_seq(items: Sequence, n_seq: int = 3, border: bool = True):
    padded = reflect_pad(items, n_seq) if border else list(items)
    return windows(padded, n_seq), padded


def gene_seq_nsf(labels, n_seq: int = 3, border: bool = True, dist: int = 7):
    lab = [int(v) for v in np.asarray(labels).squeeze().tolist()]
    if border:
        lab = reflect_pad(lab, n_seq)
    pre, sub = blurry_indices(lab, dist)
    return windows(pre, n_seq), windows(sub, n_seq)


def frame_number(path: str) -> int:
    return int(os.path.splitext(os.path.basename(path))[0])


def assemble_windows(frames: Sequence[str], labels, n_seq: int = 3, border: bool = True, max_gap: int = 7,
                     number: Callable[[str], int] = frame_number) -> List[dict]:
    """One entry per output frame: the n_seq window files, the two reference files, and whether the harness zeroes
    each reference (|frame gap| > 7 measured from the LAST window frame) — `zero_pre` is the model's routing flag."""
    seqs, padded = gene_seq(frames, n_seq, border)
    pre_w, sub_w = gene_seq_nsf(labels, n_seq, border)
    out = []
    for win, pw, sw in zip(seqs, pre_w, sub_w):
        pre_f, sub_f = padded[pw[0]], padded[sw[n_seq - 1]]
        ref_no = number(win[n_seq - 1])
        out.append({"name": os.path.splitext(os.path.basename(win[n_seq // 2]))[0], "window": list(win), "pre": pre_f, "sub": sub_f,
                    "zero_pre": abs(ref_no - number(pre_f)) > max_gap, "zero_sub": abs(ref_no - number(sub_f)) > max_gap})
    return out


def numpy2tensor(images: Sequence[np.ndarray], rgb_range: float = 1.0) -> torch.Tensor:
    """uint8 HWC images -> [1, n, 3, H, W] float32 in [0, rgb_range]."""
    ts = [torch.from_numpy(np.ascontiguousarray(np.asarray(im).astype("float64").transpose(2, 0, 1))).float().mul_(rgb_range / 255)
          for im in images]
    return torch.stack(ts).unsqueeze(0)


def numpy2tensor_device(images: Sequence[np.ndarray], device, rgb_range: float = 1.0) -> torch.Tensor:
    """Same values as `numpy2tensor`, but the uint8 frames cross PCIe (4x fewer bytes) and are converted on the device:
    uint8 -> float32 is exact and the one float32 multiply rounds identically."""
    u8 = torch.from_numpy(np.stack([np.asarray(im) for im in images])).to(device, non_blocking=True)      # [n, H, W, 3]
    return u8.permute(0, 3, 1, 2).float().mul_(rgb_range / 255).unsqueeze(0).contiguous()


def uint8_frames_to_input(frames: Sequence[torch.Tensor], rgb_range: float = 1.0) -> torch.Tensor:
    """Device-side `numpy2tensor`: uint8 [H,W,3] frames already on the device -> [1, n, 3, H, W] float32 (same values)."""
    return torch.stack(list(frames)).permute(0, 3, 1, 2).float().mul_(rgb_range / 255).unsqueeze(0).contiguous()


def tensor2numpy(t: torch.Tensor, rgb_range: float = 1.0) -> np.ndarray:
    img = t.mul(255 / rgb_range).clamp(0, 255).round()[0]
    return np.transpose(img.cpu().numpy(), (1, 2, 0)).astype(np.uint8)


def calc_psnr(a: np.ndarray, b: np.ndarray) -> float:
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
