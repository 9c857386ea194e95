"""Deterministic synthetic weights, keyed on the state_dict entry name.

The authors' checkpoints are not available offline (reference README.md:45-49), so both the
oracle and the HIP path are driven with weights produced by this generator.  It is applied
through ``load_state_dict`` so the same numbers reach the reference (when golden vectors are
made), the oracle restatement and the HIP module; no 123 MB weight file has to travel.

Every entry is drawn from ``numpy.random.RandomState(crc32(name) ^ seed)`` so the values do
not depend on iteration order, torch version or device.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def synth_tensor(name: str, ref: torch.Tensor, seed: int = 0) -> torch.Tensor:
    """Value for one state_dict entry; ``ref`` supplies shape/dtype (and value for int buffers)."""
    shape = tuple(ref.shape)
    if not ref.dtype.is_floating_point:
        return ref.clone()            # relative_position_index, num_batches_tracked
    if name.endswith("attn_mask"):
        return ref.clone()            # 0 / -100 buffer tied to patch_size (reference swinir.py:215-236)
    r = _rs(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":
        v = 0.6 + 0.8 * r.rand(*shape)
    elif leaf == "running_mean":
        v = 0.1 * r.randn(*shape)
    elif leaf == "relative_position_bias_table":
        v = 0.3 * r.randn(*shape)
    elif leaf == "weight" and len(shape) == 1:
        # LayerNorm / BatchNorm scale.  The ResBlock gates have no sigmoid (reference block.py:65-67), so
        # x1*(G1+G2) is quadratic in x1; trained weights keep it bounded, random ones need a small BN scale.
        v = (0.25 if ".te." in name else 1.0) + 0.1 * r.randn(*shape)
    elif name == "recons_net.outBlock.3.bias":
        v = 0.5 + 0.02 * r.randn(*shape)          # keeps the synthetic output mostly inside [0,1] for the PSNR metric
    elif leaf == "bias":
        v = 0.05 * r.randn(*shape)
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        gain = 0.5 if ".te." in name else (0.08 if name == "recons_net.outBlock.3.weight" else 1.0)
        v = r.randn(*shape) * (gain / np.sqrt(fan_in))
    else:
        v = 0.1 * r.randn(*shape)
    return torch.from_numpy(np.asarray(v, dtype=np.float32)).to(ref.dtype)


def state_dict_template(path: str | None = None) -> dict:
    """Zero tensors with the reference's 1020 state_dict names/shapes/dtypes.

    The inventory file was written by tests/golden/make_golden.py from the reference module itself;
    int buffers (``relative_position_index``) and ``attn_mask`` are rebuilt from their defining formulas
    (reference model/swinir.py:92-102, :215-236) because the generator passes them through unchanged.
    """
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "state_dict_keys.txt")
    out = {}
    for line in open(path):
        k, shp, dt = line.rstrip("\n").split("\t")
        shape = tuple(int(x) for x in shp.split(",")) if shp else ()
        out[k] = torch.zeros(shape, dtype=getattr(torch, dt))
    for k in out:
        if k.endswith("relative_position_index"):
            out[k] = _rel_pos_index(5)
        elif k.endswith("attn_mask"):
            out[k] = _shift_mask(50, 50, 5, 2)
    return out


def _rel_pos_index(ws: int) -> torch.Tensor:
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_mask(h: int, w: int, ws: int, shift: int) -> torch.Tensor:
    def region(n):
        r = torch.zeros(n, dtype=torch.long)
        r[n - ws:n - shift] = 1
        r[n - shift:] = 2
        return r
    reg = (3 * region(h)[:, None] + region(w)[None, :]).view(h // ws, ws, w // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    d = reg[:, None, :] - reg[:, :, None]
    return torch.where(d != 0, torch.tensor(-100.0), torch.tensor(0.0))


def synth_state_dict(template: dict, seed: int = 0) -> dict:
    """Fill every entry of ``template`` (a state_dict) with synthetic values."""
    return {k: synth_tensor(k, v, seed) for k, v in template.items()}


def synth_frames(b: int, h: int, w: int, seed: int = 1234, zero_ref: tuple = ()) -> torch.Tensor:
    """Image-like 5-frame windows ``[b,5,3,h,w]`` in [0,1].

    Frames are smooth random fields plus a little per-frame motion so the SearchTransfer
    correlation has structure; samples listed in ``zero_ref`` get frame 3 zeroed, which routes
    them to the no-reference branch (reference model/speinet.py:70-73,150-168).
    """
    r = np.random.RandomState(seed)
    out = np.empty((b, 5, 3, h, w), dtype=np.float32)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    for bi in range(b):
        base = np.zeros((3, h, w), dtype=np.float32)
        for _ in range(12):
            fy, fx = r.uniform(0.02, 0.45, size=2)
            ph = r.uniform(0, 2 * np.pi, size=3)
            amp = r.uniform(0.05, 0.25)
            for c in range(3):
                base[c] += amp * np.sin(fy * yy + fx * xx + ph[c])
        base = 0.5 + base / 2.5
        for f in range(5):
            noise = 0.01 * r.randn(3, h, w).astype(np.float32)
            shift = np.roll(base, (f - 2, 2 - f), axis=(1, 2))
            out[bi, f] = np.clip(shift + noise, 0.0, 1.0)
            out[bi, f, :, :2, :3] = 0.0      # a few exact zeros: exercises 0/0 -> NaN -> 0 in the RL prior
        if bi in zero_ref:
            out[bi, 3] = 0.0
    return torch.from_numpy(out)


def synth_frames_edges(b: int, h: int, w: int, seed: int = 1700, zero_ref: tuple = ()):
    """A second, edge-dominated content class (the shape of a deblurring workload rather than of a smooth field):
    a sharp scene of random flat rectangles and discs over a gentle gradient plus fine texture; frames 0..2 are
    that scene under per-frame translation and a horizontal box blur (motion-blur stand-in, 7..13 taps), frame 3 a
    sharp neighbour, frame 4 the sharp reference.  Returns (x [b,5,3,h,w] in [0,1], gt [b,3,h,w]): gt is the sharp
    scene at the middle frame's position, the PSNR target of the full-size golden case G17."""
    r = np.random.RandomState(seed)
    x = np.empty((b, 5, 3, h, w), dtype=np.float32)
    gt = np.empty((b, 3, h, w), dtype=np.float32)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    for bi in range(b):
        scene = np.empty((3, h, w), dtype=np.float32)
        for c in range(3):
            scene[c] = 0.35 + 0.3 * (yy / h) * r.uniform(0.3, 1.0) + 0.2 * (xx / w) * r.uniform(0.3, 1.0)
        for _ in range(60):
            col = r.uniform(0.05, 0.95, size=3).astype(np.float32)
            cy, cx = r.uniform(0, h), r.uniform(0, w)
            ry, rx = r.uniform(h * 0.02, h * 0.2), r.uniform(w * 0.02, w * 0.2)
            if r.rand() < 0.5:
                m = (np.abs(yy - cy) < ry) & (np.abs(xx - cx) < rx)
            else:
                m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
            scene[:, m] = col[:, None]
        tex = 0.03 * np.sin(0.9 * xx + 0.4 * yy) * np.sin(0.23 * yy - 0.11 * xx)
        scene = np.clip(scene + tex[None], 0.0, 1.0).astype(np.float32)
        for f in range(5):
            dy, dx = (f - 1) * 2, (1 - f) * 3
            moved = np.roll(scene, (dy, dx), axis=(1, 2))
            if f < 3:
                taps = 7 + 2 * int(r.randint(0, 4))
                acc = np.zeros_like(moved)
                for t in range(taps):
                    acc += np.roll(moved, t - taps // 2, axis=2)
                moved = acc / taps
            x[bi, f] = np.clip(moved + 0.004 * r.randn(3, h, w).astype(np.float32), 0.0, 1.0)
        gt[bi] = scene                              # frame 1 has dy = dx = 0
        if bi in zero_ref:
            x[bi, 3] = 0.0
    return torch.from_numpy(x), torch.from_numpy(gt)
