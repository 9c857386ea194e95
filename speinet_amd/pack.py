"""Checkpoint -> kernel-friendly weight layouts (done once per load, never stored in the checkpoint).

Layouts consumed by the C-ABI (include/speinet_hip.h):
  conv / linear weights   [tap][Cout][Cin] fp32  (tap = ky*k + kx; Linear is one tap)
  ConvTranspose2d         same, from the [Cin][Cout][k][k] checkpoint layout
  ResBlock gates          SE matrices as stored; 2->1 channel gate convs flattened; eval BatchNorm(1) folded to
                          (scale, shift)
  Swin block              LayerNorm affine folded into the following linear (exact algebra, done in float64):
                            q  = LN(y) Wq^T + bq  ->  yhat (scale*Wq*gamma)^T + scale*(Wq beta + bq)
                            kv = LN(x) Wkv^T + bkv, fc1 likewise; relative position bias gathered to [8][25][25]
                          (reference model/swinir.py:105-108,124-134,244-245,279)
"""
from __future__ import annotations

from typing import Dict

import torch

SD = Dict[str, torch.Tensor]


class GemmW:
    """Marker: a [tap][Cout][Cin] (or [Cout][Cin]) GEMM weight still on the host side of packing."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t if t.dim() == 3 else t.reshape(1, *t.shape)


F32, BF16, F16 = 0, 1, 2          # include/speinet_hip.h SPEI_F32 / SPEI_BF16 / SPEI_F16
LP_DTYPE = {BF16: torch.bfloat16, F16: torch.float16}


def fmt_of(dtype) -> int:
    return {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[dtype]


def _frag(w: torch.Tensor) -> torch.Tensor:
    """[tap][N][K] 16-bit -> MFMA fragment order for the slab kernels: [n-tile][tap][k-step of 16][lane = h*32 + n%32][8],
    element (t, n, k) with k = 16*ks + 8*h + j  ->  one coalesced 1 KiB load per B fragment."""
    t, n, k = w.shape
    return w.view(t, n // 32, 32, k // 16, 2, 8).permute(1, 0, 3, 4, 2, 5).contiguous()


class PackedW:
    """Device copies of one GEMM weight: f32 for the exact path; bf16 hi / lo (= bf16(w - hi)) plain and in fragment
    order for the bf16 / bf16x3 modes; half in fragment order for the f16 mode (`frag(fmt)`)."""
    __slots__ = ("f32", "hi", "lo", "shape", "fhi", "flo", "fh16", "_ct")

    def __init__(self, t: torch.Tensor, device):
        self.f32 = t.detach().to(device=device, dtype=torch.float32).contiguous()
        self.hi = self.f32.to(torch.bfloat16)
        self.lo = (self.f32 - self.hi.float()).to(torch.bfloat16)
        self.shape = tuple(self.f32.shape)
        t, n, k = self.shape
        self.fhi = self.flo = self.fh16 = None
        self._ct = {}
        if n % 32 == 0 and k % 32 == 0:
            self.fhi, self.flo = _frag(self.hi), _frag(self.lo)
            self.fh16 = _frag(self.f32.to(torch.float16))

    def frag(self, fmt: int):
        return self.fh16 if fmt == F16 else self.fhi

    def convT_class_frags(self, fmt: int = BF16) -> dict:
        """Fragment-ordered weights of the four output-parity classes of a 3x3 stride-2 transposed conv (this pack holds
        [ky*3+kx][Cout][Cin]): out[2y+py][2x+px] = sum_taps in[y+dy][x+dx] W[ky][kx] with parity 0 -> k = 1 (d = 0), parity
        1 -> k = 0 (d = 1), k = 2 (d = 0) — the tap order spei_convt2_slab16 assumes."""
        if fmt not in self._ct:
            assert self.shape[0] == 9
            self._ct[fmt] = self._class_frags(self.f32.to(LP_DTYPE[fmt]))
        return self._ct[fmt]

    @staticmethod
    def _class_frags(src: torch.Tensor) -> dict:
        ct = {}
        for py in (0, 1):
            for px in (0, 1):
                kys, kxs = ([0, 2] if py else [1]), ([0, 2] if px else [1])
                ct[(py, px)] = _frag(torch.stack([src[ky * 3 + kx] for ky in kys for kx in kxs]))
        return ct

    def convT_class_frags_split(self):
        """(hi, lo) class fragments for the split (bf16x3) form, spei_convt2_slab16x3, + the two device arrays of four pointers it takes."""
        if "split" not in self._ct:
            assert self.shape[0] == 9
            hi, lo = self._class_frags(self.hi), self._class_frags(self.lo)
            order = [(0, 0), (0, 1), (1, 0), (1, 1)]
            import ctypes as C
            ph = (C.c_void_p * 4)(*[hi[k].data_ptr() for k in order])
            pl = (C.c_void_p * 4)(*[lo[k].data_ptr() for k in order])
            self._ct["split"] = (hi, lo, ph, pl)
        return self._ct["split"]


def conv_w(w: torch.Tensor) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci).contiguous()


def convT_w(w: torch.Tensor) -> torch.Tensor:
    ci, co, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw, co, ci).contiguous()


def G(t: torch.Tensor) -> GemmW:
    return GemmW(t)


def _bn_fold(sd: SD, p: str) -> torch.Tensor:
    scale = sd[p + "weight"].double() / torch.sqrt(sd[p + "running_var"].double() + 1e-5)
    shift = sd[p + "bias"].double() - sd[p + "running_mean"].double() * scale
    return torch.stack((scale.reshape(()), shift.reshape(()))).float()


def resblock(sd: SD, p: str) -> dict:
    return {
        "w1": G(conv_w(sd[p + "main.0.main.0.weight"])), "b1": sd[p + "main.0.main.0.bias"],
        "w2": G(conv_w(sd[p + "main.1.main.0.weight"])), "b2": sd[p + "main.1.main.0.bias"],
        "se_w1": sd[p + "se.fc.0.weight"], "se_b1": sd[p + "se.fc.0.bias"],
        "se_w2": sd[p + "se.fc.2.weight"], "se_b2": sd[p + "se.fc.2.bias"],
        "cw_w": sd[p + "te.cw.conv.conv.weight"].reshape(-1), "cw_bn": _bn_fold(sd, p + "te.cw.conv.bn."),
        "hc_w": sd[p + "te.hc.conv.conv.weight"].reshape(-1), "hc_bn": _bn_fold(sd, p + "te.hc.conv.bn."),
    }


def rel_pos_index(ws: int) -> torch.Tensor:
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swin_block(sd: SD, p: str, heads: int, ws: int) -> dict:
    g1, b1 = sd[p + "norm1.weight"].double(), sd[p + "norm1.bias"].double()
    g2, b2 = sd[p + "norm2.weight"].double(), sd[p + "norm2.bias"].double()
    wq, bq = sd[p + "attn.qkv_y.weight"].double(), sd[p + "attn.qkv_y.bias"].double()
    wkv, bkv = sd[p + "attn.qkv_x.weight"].double(), sd[p + "attn.qkv_x.bias"].double()
    w1, bb1 = sd[p + "mlp.fc1.weight"].double(), sd[p + "mlp.fc1.bias"].double()
    dim = wq.shape[1]
    scale = (dim // heads) ** -0.5
    n = ws * ws
    idx = rel_pos_index(ws).reshape(-1).to(sd[p + "attn.relative_position_bias_table"].device)
    relb = sd[p + "attn.relative_position_bias_table"][idx].reshape(n, n, heads).permute(2, 0, 1).contiguous()
    return {
        "wq": G((scale * wq * g1[None, :]).float().contiguous()), "bq": (scale * (wq @ b1 + bq)).float(),
        "wkv": G((wkv * g1[None, :]).float().contiguous()), "bkv": (wkv @ b1 + bkv).float(),
        "wproj": G(sd[p + "attn.proj.weight"].contiguous()), "bproj": sd[p + "attn.proj.bias"],
        "w1": G((w1 * g2[None, :]).float().contiguous()), "b1": (w1 @ b2 + bb1).float(),
        "w2": G(sd[p + "mlp.fc2.weight"].contiguous()), "b2": sd[p + "mlp.fc2.bias"],
        "relbias": relb.float(),
    }


def _to_device(o, device):
    if isinstance(o, GemmW):
        return PackedW(o.t, device)
    if torch.is_tensor(o):
        return o.detach().to(device=device, dtype=torch.float32).contiguous()
    if isinstance(o, dict):
        return {k: _to_device(v, device) for k, v in o.items()}
    if isinstance(o, list):
        return [_to_device(v, device) for v in o]
    return o


def pack_swint(sd: SD, cfg, device) -> dict:
    """Packed tensors of the `swint` variant (reference model/swint.py): the same encoder / decoder stacks and SwinIR, one
    1x1 fusion conv named `conv`, no SearchTransfer glue."""
    out = _pack_common(sd, cfg)
    out["conv"] = {"w": G(conv_w(sd["conv.weight"])), "b": sd["conv.bias"]}
    return _to_device(out, device)


def pack_all(sd: SD, cfg, device) -> dict:
    """Every packed tensor the SPEINet forward pass needs, on `device`, fp32 contiguous."""
    out = _pack_common(sd, cfg)
    for name in ("conv_lv1", "conv_lv2", "conv_lv3", "fusion", "search1", "search2", "search3", "search13", "search33", "search43"):
        out[name] = {"w": G(conv_w(sd[name + ".weight"])), "b": sd[name + ".bias"]}
    for name in ("search1", "search2"):
        out["SelfTransfer." + name] = {"w": G(conv_w(sd[f"SelfTransfer.{name}.weight"])), "b": sd[f"SelfTransfer.{name}.bias"]}
    return _to_device(out, device)


def _pack_common(sd: SD, cfg) -> dict:
    """recons_net stacks + SwinIR (shared by model/speinet.py and model/swint.py), still on the host side."""
    out: dict = {}
    nrb = cfg.n_resblock

    def stage(name, head_idx, nblocks, first):
        d = {}
        p = f"recons_net.{name}."
        if head_idx is not None:
            hw = conv_w(sd[f"{p}{head_idx}.0.weight"])
            d["head_w"] = hw if name == "inBlock" else G(hw)     # the 3-channel head has its own direct kernel
            d["head_b"] = sd[f"{p}{head_idx}.0.bias"]
        d["blocks"] = [resblock(sd, f"{p}{first + i}.") for i in range(nblocks)]
        return d

    out["inBlock"] = stage("inBlock", 0, nrb, 1)
    out["encoder_first"] = stage("encoder_first", 0, nrb, 1)
    out["encoder_second"] = stage("encoder_second", 0, nrb, 1)
    for name in ("decoder_second", "decoder_first"):
        d = stage(name, None, nrb, 0)
        d["tail_w"] = G(convT_w(sd[f"recons_net.{name}.{nrb}.0.weight"]))
        d["tail_b"] = sd[f"recons_net.{name}.{nrb}.0.bias"]
        out[name] = d
    d = stage("outBlock", None, nrb, 0)
    d["tail_w"] = conv_w(sd[f"recons_net.outBlock.{nrb}.weight"])
    d["tail_b"] = sd[f"recons_net.outBlock.{nrb}.bias"]
    # the same 32 -> 3 conv zero-padded to 32 output channels: the bf16 mode runs it on the slab kernel
    d["tail_w32"] = G(torch.cat((d["tail_w"], torch.zeros(25, 29, d["tail_w"].shape[2], dtype=d["tail_w"].dtype, device=d["tail_w"].device)), 1))
    d["tail_b32"] = torch.cat((d["tail_b"], torch.zeros(29, dtype=d["tail_b"].dtype, device=d["tail_b"].device)))
    out["outBlock"] = d

    r = float(cfg.rgb_range)
    sw = {
        "conv_first_w": G(conv_w(sd["swin.conv_first.weight"] * r)), "conv_first_b": sd["swin.conv_first.bias"],
        "pe_g": sd["swin.patch_embed.norm.weight"], "pe_b": sd["swin.patch_embed.norm.bias"],
        "norm_g": sd["swin.norm.weight"], "norm_b": sd["swin.norm.bias"],
        "cab_w": G(conv_w(sd["swin.conv_after_body.weight"])), "cab_b": sd["swin.conv_after_body.bias"],
        "conv_last_w": G(conv_w(sd["swin.conv_last.weight"] / r)), "conv_last_b": sd["swin.conv_last.bias"] / r,
        "layers": [],
    }
    for li, depth in enumerate(cfg.depths):
        p = f"swin.layers.{li}."
        sw["layers"].append({
            "blocks": [swin_block(sd, f"{p}residual_group.blocks.{bi}.", cfg.num_heads[li], cfg.window_size) for bi in range(depth)],
            "conv_w": G(conv_w(sd[p + "conv.weight"])), "conv_b": sd[p + "conv.bias"],
        })
    out["swin"] = sw
    return out
