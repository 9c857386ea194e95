"""Drop-in for the reference's ``model/swint.py`` (the model `main_swint.py` / `trainer/trainer_swint.py` build with
``--model swint``): same class name, constructor, ``make_model(args)`` factory and ``state_dict`` layout — ``swin.*``,
``recons_net.*`` and the 1x1 fusion conv ``conv.{weight,bias}`` — with ``forward`` on the HIP path.

    forward(x [B, >= n_sequence, 3, H, W]) -> [B, 3, H, W]                     (reference model/swint.py:51-67)
        f_mid   = enc(x[:, n//2]);   f_i = swin(f_mid, enc(x[:, i]))  for the other frames
        n == 1:   f = f_mid + swin(f_mid, f_mid)
        out     = outBlock(decoder_first(decoder_second(conv(cat(f_mid, f_i...)))))

It is a strict subset of the SPEINet kernels (no RL prior, no SearchTransfer): same arithmetic modes, same packing, same
engine pieces.  In train() mode, or whenever autograd is recording, `forward` builds the differentiable fp32 graph of
speinet_amd.train (HIP forward AND backward kernels, BatchNorm(1) batch statistics, DropPath), so the reference trainer's
`loss.backward(); optimizer.step()` works on this module unchanged.  No CPU path, as for speinet_amd.speinet.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, engine, ops, pack
from .speinet import _Recons, _SwinIR, _wants_autograd, default_args


class SPEINet(nn.Module):
    """Same signature as reference model/swint.py:19-23 (the class there is also called SPEINet)."""

    def __init__(self, in_channels=3, n_sequence=3, out_channels=3, n_resblock=3, n_feat=32,
                 load_flow_net=False, load_recons_net=False, flow_pretrain_fn='', recons_pretrain_fn='',
                 is_mask_filter=False, device='cuda', args=None):
        super().__init__()
        if args is None:
            args = default_args()
        if in_channels != 3 or out_channels != 3 or n_feat != 32 or n_sequence not in (1, 3, 5):
            raise ValueError("speinet_amd.swint builds 3 colours, n_feat=32, n_sequence in {1, 3, 5}")
        if args.window_size != 5 or args.embed_dim != 256 or any(h != 8 for h in args.num_heads):
            raise ValueError("speinet_amd kernels are built for window 5, embed_dim 256, 8 heads")
        if getattr(args, "resi_connection", "1conv") != "1conv":
            raise ValueError("only resi_connection='1conv' is built")
        self.n_sequence = n_sequence
        self.device = device
        self.is_mask_filter = is_mask_filter
        self.cfg = SimpleNamespace(n_sequence=n_sequence, n_feat=n_feat, n_resblock=n_resblock, window_size=args.window_size,
                                   embed_dim=args.embed_dim, depths=tuple(args.depths), num_heads=tuple(args.num_heads),
                                   mlp_ratio=args.mlp_ratio, rgb_range=float(args.rgb_range), patch_size=args.patch_size)
        self.swin = _SwinIR(n_feat * 4, args.patch_size // 4, args.window_size, list(args.depths), args.embed_dim,
                            list(args.num_heads), args.mlp_ratio)
        self.recons_net = _Recons(in_channels, out_channels, n_resblock, n_feat)
        self.conv = nn.Conv2d(n_feat * 4 * n_sequence, n_feat * 4, 1)
        if load_recons_net:
            self.recons_net.load_state_dict(torch.load(recons_pretrain_fn, weights_only=True))
        self._packed = {}
        self.precision = os.environ.get("SPEINET_PRECISION", "f32")
        self.streams = int(os.environ.get("SPEINET_STREAMS", "1"))
        self.knobs = {}
        self.train_precision = os.environ.get("SPEINET_TRAIN_PRECISION", "f32")   # GEMMs of the training graph: "f32" | "bf16x3" (train.py)
        self._side_streams = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self) -> None:
        """Drop the packed weights (automatic after load_state_dict / .to(); call it after in-place parameter edits).  Launches on the
        side streams may still be reading them by raw pointer: the devices that hold any are synchronised first."""
        for key in list(self._packed):
            torch.cuda.synchronize(torch.device(key))
        self._packed = {}
        from .train import drop_split_cache
        drop_split_cache(self)           # the training graph's packed bf16 halves, kept on the parameters (train._split_frags)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_packed()
        return out

    def _pack(self, device):
        key = str(torch.device(device))
        if key not in self._packed:
            self._packed[key] = pack.pack_swint(self.state_dict(), self.cfg, device)
        return self._packed[key]

    def forward(self, x: torch.Tensor, profile: Optional[dict] = None, drop_path_scales: Optional[list] = None) -> torch.Tensor:
        if x.dim() != 5 or x.shape[1] < self.n_sequence or x.shape[2] != 3:
            raise ValueError(f"expected [B, >= {self.n_sequence}, 3, H, W], got {tuple(x.shape)}")
        h, w = x.shape[-2:]
        if h % 20 or w % 20:
            raise ValueError(f"H and W must be multiples of 20 (two stride-2 stages, then 5x5 windows); got {h}x{w}")
        if not x.is_cuda:
            raise RuntimeError("speinet_amd.swint runs on MI355X only (HIP kernels); there is no CPU path")
        _lib.lib()
        if _wants_autograd(self, x):
            # the differentiable graph (fp32, HIP forward and backward kernels): train() mode runs BatchNorm(1) on batch
            # statistics and DropPath, as the reference module does under trainer/trainer_swint.py:27,39; eval() with
            # `autograd = True` (or an input that requires grad) is the same graph with running statistics and no DropPath.
            # An optimizer step may follow either way: the packed inference weights are stale from here on
            from . import train
            self.invalidate_packed()
            return train.forward_swint(self, x, scales=drop_path_scales)
        with torch.cuda.device(x.device):
            ctx = ops.Ctx(self.precision, "top2", device=x.device, profile=profile, **self.knobs)
            x = x.contiguous().float()
            P = self._pack(x.device)
            n = max(1, int(self.streams)) - 1
            key = (x.device.index, n)
            if key not in self._side_streams:
                self._side_streams[key] = [torch.cuda.Stream(device=x.device) for _ in range(n)]
            out = torch.empty(x.shape[0], 3, h, w, device=x.device, dtype=torch.float32)
            for b in range(x.shape[0]):
                engine.forward_swint(ctx, x[b], P, self.n_sequence, out[b], self._side_streams[key])
            return out


def make_model(args):
    """reference model/swint.py:8-16."""
    device = 'cpu' if getattr(args, "cpu", False) else 'cuda'
    return SPEINet(in_channels=args.n_colors, n_sequence=args.n_sequence, out_channels=args.n_colors,
                   n_resblock=args.n_resblock, n_feat=args.n_feat, load_recons_net=False, recons_pretrain_fn='',
                   is_mask_filter=True, device=device, args=args)
