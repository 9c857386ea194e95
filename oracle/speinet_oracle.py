"""ORACLE — test infrastructure only.  CPU restatement (PyTorch fp32, functional) of the reference
SPEINet per-sequence forward pass (and, inside ``train_mode``, of the training graph: torch autograd over the same functions).
Nothing under ``speinet_amd/`` or ``tools/`` may import this file; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` legs (inference and ``--train``) use it, as the checker / the CPU baseline.

Pinning: checked in ``tests/test_oracle_golden.py`` against golden vectors produced by importing
the reference itself in the build container (``tests/golden/make_golden.py``; recipe in SURVEY.md §8c).
The reference has no tests/fixtures of its own so those vectors *are* the pin.

All ``file:line`` citations are relative to the reference tree (yangt1013/SPEINet @ 2025-02-11).
Weights come in as a plain ``state_dict`` mapping (reference key names).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class Cfg:
    """Hyper-parameters the reference reads from ``args`` (model/speinet.py:40-49, option/template.py:3-21)."""
    n_sequence: int = 3
    n_feat: int = 32
    n_resblock: int = 3
    window_size: int = 5
    embed_dim: int = 256
    depths: Tuple[int, ...] = (6, 6, 6, 6, 6, 6)
    num_heads: Tuple[int, ...] = (8, 8, 8, 8, 8, 8)
    mlp_ratio: float = 2
    rgb_range: float = 1.0
    patch_size: int = 200


# --------------------------------------------------------------------------------------------
# K1  Richardson-Lucy style prior  (model/rcl.py:18-51)
# --------------------------------------------------------------------------------------------
def rl_prior(img: torch.Tensor, iters: int, lam: float = 0.01) -> torch.Tensor:
    """``r_l_per_channel(img, create_blur_kernel(), iters, lam)``  (model/rcl.py:22-51).

    Per channel: b = box5(d) (zero pad 2, weights 1/25); c = I / b with NaN -> 0 and negatives -> 0
    (inf is NOT handled, rcl.py:38-40); d <- c * (d + lam * lap4(d)) with lap4 = [[0,-1,0],[-1,4,-1],[0,-1,0]]
    zero padded.
    """
    b_, c_, h, w = img.shape
    x = img.reshape(b_ * c_, 1, h, w)
    box = torch.ones(1, 1, 5, 5, dtype=img.dtype) / 25.0                 # (the reference builds both in fp32, its only dtype)
    lap = torch.tensor([[0, -1, 0], [-1, 4, -1], [0, -1, 0]], dtype=img.dtype).view(1, 1, 3, 3)
    d = x.clone()
    for _ in range(iters):
        blurred = F.conv2d(d, box, padding=2)
        corr = x / blurred
        corr = torch.where(corr != corr, torch.zeros_like(corr), corr)
        corr = torch.where(corr < 0, torch.zeros_like(corr), corr)
        d = corr * (d + lam * F.conv2d(d, lap, padding=1))
    return d.reshape(b_, c_, h, w)


# --------------------------------------------------------------------------------------------
# K2/K3  ResBlock = conv5-relu-conv5, SE + TripletAttention gates, skip  (model/block.py:8-140)
# --------------------------------------------------------------------------------------------
class _TrainState:
    """Train-mode switches of the restatement (used by the training-step checks only; default = the eval graph):
    bn_batch — BatchNorm2d(1) normalises with the statistics of the batch (nn.BatchNorm2d in train(), model/block.py:56);
    drop     — iterator over the DropPath factors in call order, one (attention [B], mlp [B]) pair or None per Swin block
               (timm.models.layers.DropPath as used at model/swinir.py:203,278-279)."""
    bn_batch = False
    drop = None


_TRAIN = _TrainState()


class train_mode:
    """``with train_mode(drop_scales):`` — evaluate the restatement as the reference module computes in train(): batch
    statistics in the gates' BatchNorm (the running buffers are not touched: the restatement is functional) and the given
    DropPath factors ([call][block] -> None | (attn [B], mlp [B]); None = no DropPath).  With tensors of ``sd`` that require
    grad, ``loss.backward()`` then gives the training gradients."""

    def __init__(self, drop_scales=None):
        self.flat = None if drop_scales is None else [pair for call in drop_scales for pair in call]

    def __enter__(self):
        _TRAIN.bn_batch, _TRAIN.drop = True, (iter(self.flat) if self.flat is not None else None)
        return self

    def __exit__(self, *exc):
        left = list(_TRAIN.drop) if _TRAIN.drop is not None else []
        _TRAIN.bn_batch, _TRAIN.drop = False, None
        assert exc[0] is not None or not left, f"{len(left)} DropPath factors were not consumed"
        return False


def _bn1_eval(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """BatchNorm2d(1), eps 1e-5 (model/block.py:56): running statistics (eval), or the batch's (inside ``train_mode``)."""
    if _TRAIN.bn_batch:
        m, v = x.mean(), x.var(unbiased=False)
        return (x - m) / torch.sqrt(v + 1e-5) * sd[p + "weight"].view(1, -1, 1, 1) + sd[p + "bias"].view(1, -1, 1, 1)
    return (x - sd[p + "running_mean"].view(1, -1, 1, 1)) / torch.sqrt(sd[p + "running_var"].view(1, -1, 1, 1) + 1e-5) \
        * sd[p + "weight"].view(1, -1, 1, 1) + sd[p + "bias"].view(1, -1, 1, 1)


def resblock_gates(x1: torch.Tensor, sd: SD, p: str):
    """Returns (s[B,C,1,1], G1[B,1,H,C]->as [B,C,H,1], G2[B,C,1,W]) so that ResBlock = x + x1*(s+G1+G2).

    SE: model/block.py:8-24.  cw gate: ZPool over W then 7x7 conv on the (H,C) plane + BN, no sigmoid
    (block.py:75-84 with relu=False => block.py:65-67 skips sigmoid).  hc gate: ZPool over H, 5x5 conv on
    the (C,W) plane + BN (block.py:86-96).  TripletAttention sums the two branches (block.py:116-119).
    """
    b, c, h, w = x1.shape
    y = x1.mean(dim=(2, 3))
    y = F.relu(F.linear(y, sd[p + "se.fc.0.weight"], sd[p + "se.fc.0.bias"]))
    s = torch.sigmoid(F.linear(y, sd[p + "se.fc.2.weight"], sd[p + "se.fc.2.bias"])).view(b, c, 1, 1)
    # cw: x.permute(0,3,2,1) -> [B,W,H,C]; ZPool over dim 1 (= W)
    xp = x1.permute(0, 3, 2, 1)
    z = torch.cat((xp.max(dim=1)[0].unsqueeze(1), xp.mean(dim=1).unsqueeze(1)), dim=1)      # [B,2,H,C]
    g1 = _bn1_eval(F.conv2d(z, sd[p + "te.cw.conv.conv.weight"], padding=3), sd, p + "te.cw.conv.bn.")  # [B,1,H,C]
    g1 = g1.permute(0, 3, 2, 1)                                                              # [B,C,H,1]
    # hc: x.permute(0,2,1,3) -> [B,H,C,W]; ZPool over dim 1 (= H)
    xp = x1.permute(0, 2, 1, 3)
    z = torch.cat((xp.max(dim=1)[0].unsqueeze(1), xp.mean(dim=1).unsqueeze(1)), dim=1)      # [B,2,C,W]
    g2 = _bn1_eval(F.conv2d(z, sd[p + "te.hc.conv.conv.weight"], padding=2), sd, p + "te.hc.conv.bn.")  # [B,1,C,W]
    g2 = g2.permute(0, 2, 1, 3)                                                              # [B,C,1,W]
    return s, g1, g2


def resblock(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """model/block.py:127-140: ``x + SE(x1) + TE(x1)`` with ``x1 = conv5(relu(conv5(x)))``."""
    x1 = F.relu(F.conv2d(x, sd[p + "main.0.main.0.weight"], sd[p + "main.0.main.0.bias"], padding=2))
    x1 = F.conv2d(x1, sd[p + "main.1.main.0.weight"], sd[p + "main.1.main.0.bias"], padding=2)
    s, g1, g2 = resblock_gates(x1, sd, p)
    return x + x1 * (s + g1 + g2)


def _stage(x: torch.Tensor, sd: SD, p: str, stride: int, n_resblock: int) -> torch.Tensor:
    """conv5(stride)+ReLU then n ResBlocks: inBlock / encoder_first / encoder_second
    (model/recons_video_ori.py:26-56)."""
    x = F.relu(F.conv2d(x, sd[p + "0.0.weight"], sd[p + "0.0.bias"], stride=stride, padding=2))
    for i in range(n_resblock):
        x = resblock(x, sd, f"{p}{i + 1}.")
    return x


def in_block(x, sd, cfg: Cfg):
    return _stage(x, sd, "recons_net.inBlock.", 1, cfg.n_resblock)


def encoder_first(x, sd, cfg: Cfg):
    return _stage(x, sd, "recons_net.encoder_first.", 2, cfg.n_resblock)


def encoder_second(x, sd, cfg: Cfg):
    return _stage(x, sd, "recons_net.encoder_second.", 2, cfg.n_resblock)


def enc(x, sd, cfg: Cfg):
    """``encoder_second(encoder_first(inBlock(x)))`` (model/speinet.py:82-83,130-131)."""
    return encoder_second(encoder_first(in_block(x, sd, cfg), sd, cfg), sd, cfg)


def _dec_stage(x: torch.Tensor, sd: SD, p: str, n_resblock: int) -> torch.Tensor:
    """n ResBlocks then ConvTranspose2d(3, s2, p1, op1)+ReLU (model/recons_video_ori.py:58-71)."""
    for i in range(n_resblock):
        x = resblock(x, sd, f"{p}{i}.")
    return F.relu(F.conv_transpose2d(x, sd[f"{p}{n_resblock}.0.weight"], sd[f"{p}{n_resblock}.0.bias"],
                                     stride=2, padding=1, output_padding=1))


def out_block(x: torch.Tensor, sd: SD, cfg: Cfg) -> torch.Tensor:
    """3 ResBlocks + conv5 n_feat->3, no activation (model/recons_video_ori.py:73-77)."""
    p = "recons_net.outBlock."
    for i in range(cfg.n_resblock):
        x = resblock(x, sd, f"{p}{i}.")
    return F.conv2d(x, sd[f"{p}{cfg.n_resblock}.weight"], sd[f"{p}{cfg.n_resblock}.bias"], padding=2)


# --------------------------------------------------------------------------------------------
# K7-K9  cross-window-attention SwinIR  (model/swinir.py)
# --------------------------------------------------------------------------------------------
def rel_pos_index(ws: int) -> torch.Tensor:
    """model/swinir.py:92-102."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(h: int, w: int, ws: int, shift: int) -> torch.Tensor:
    """``calculate_mask`` (model/swinir.py:215-236): [nW, ws*ws, ws*ws] of 0 / -100."""
    img = torch.zeros(1, h, w, 1)          # region ids: small integers, exact in any float type
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def window_partition(x: torch.Tensor, ws: int) -> torch.Tensor:
    """model/swinir.py:32-44.  x [B,H,W,C] -> [B*nW, ws, ws, C]."""
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, c)


def window_reverse(win: torch.Tensor, ws: int, h: int, w: int) -> torch.Tensor:
    """model/swinir.py:47-61."""
    b = int(win.shape[0] / (h * w / ws / ws))
    x = win.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, -1)


def window_attention(xw, yw, sd: SD, p: str, nh: int, ws: int, mask=None):
    """``WindowAttention.forward`` (model/swinir.py:115-149): K,V from x, Q from y."""
    b_, n, c = xw.shape
    kv = F.linear(xw, sd[p + "qkv_x.weight"], sd[p + "qkv_x.bias"]).reshape(b_, n, 2, nh, c // nh).permute(2, 0, 3, 1, 4)
    q = F.linear(yw, sd[p + "qkv_y.weight"], sd[p + "qkv_y.bias"]).reshape(b_, n, 1, nh, c // nh).permute(2, 0, 3, 1, 4)[0]
    k, v = kv[0], kv[1]
    q = q * ((c // nh) ** -0.5)
    attn = q @ k.transpose(-2, -1)
    idx = rel_pos_index(ws).view(-1)
    bias = sd[p + "relative_position_bias_table"][idx].view(n, n, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = attn.view(b_ // nw, nw, nh, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nh, n, n)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b_, n, c)
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])


def swin_block(x, y, x_size, sd: SD, p: str, nh: int, ws: int, shift: int):
    """``SwinTransformerBlock.forward`` (model/swinir.py:238-281).  x,y: [B, H*W, C] tokens."""
    h, w = x_size
    b, l, c = x.shape
    shortcut = x
    xn = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5).view(b, h, w, c)
    yn = F.layer_norm(y, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5).view(b, h, w, c)
    if shift > 0:
        xn = torch.roll(xn, shifts=(-shift, -shift), dims=(1, 2))
        yn = torch.roll(yn, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(xn, ws).view(-1, ws * ws, c)
    yw = window_partition(yn, ws).view(-1, ws * ws, c)
    mask = shift_mask(h, w, ws, shift) if shift > 0 else None
    aw = window_attention(xw, yw, sd, p + "attn.", nh, ws, mask).view(-1, ws, ws, c)
    xr = window_reverse(aw, ws, h, w)
    if shift > 0:
        xr = torch.roll(xr, shifts=(shift, shift), dims=(1, 2))
    pair = next(_TRAIN.drop) if _TRAIN.drop is not None else None          # DropPath factors of this block, or None
    branch = xr.view(b, h * w, c)
    x = shortcut + (branch if pair is None else branch * pair[0].to(branch.dtype).view(b, 1, 1))
    hmid = F.gelu(F.linear(F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5),
                           sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    branch = F.linear(hmid, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + (branch if pair is None else branch * pair[1].to(branch.dtype).view(b, 1, 1))


def rstb(x, y, x_size, sd: SD, p: str, depth: int, nh: int, ws: int):
    """``RSTB.forward`` (model/swinir.py:483-484): blocks, then 3x3 conv on the un-embedded map, + skip."""
    b, l, c = x.shape
    r = x
    for i in range(depth):
        r = swin_block(r, y, x_size, sd, f"{p}residual_group.blocks.{i}.", nh, ws, 0 if i % 2 == 0 else ws // 2)
    img = r.transpose(1, 2).reshape(b, c, x_size[0], x_size[1])
    img = F.conv2d(img, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)
    return img.flatten(2).transpose(1, 2) + x


def swin(x: torch.Tensor, y: torch.Tensor, sd: SD, cfg: Cfg, p: str = "swin.") -> torch.Tensor:
    """``SwinIR.forward`` live branch (model/swinir.py:781-810, else-branch :802-806), in_chans != 3 so mean = 0."""
    r = float(cfg.rgb_range)
    x = x * r
    y = y * r
    xf = F.conv2d(x, sd[p + "conv_first.weight"], sd[p + "conv_first.bias"], padding=1)
    yf = F.conv2d(y, sd[p + "conv_first.weight"], sd[p + "conv_first.bias"], padding=1)
    x_size = (xf.shape[2], xf.shape[3])
    c = xf.shape[1]
    xt = F.layer_norm(xf.flatten(2).transpose(1, 2), (c,), sd[p + "patch_embed.norm.weight"], sd[p + "patch_embed.norm.bias"], 1e-5)
    yt = F.layer_norm(yf.flatten(2).transpose(1, 2), (c,), sd[p + "patch_embed.norm.weight"], sd[p + "patch_embed.norm.bias"], 1e-5)
    for li, depth in enumerate(cfg.depths):
        xt = rstb(xt, yt, x_size, sd, f"{p}layers.{li}.", depth, cfg.num_heads[li], cfg.window_size)
    xt = F.layer_norm(xt, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    feat = xt.transpose(1, 2).reshape(xt.shape[0], c, x_size[0], x_size[1])
    res = F.conv2d(feat, sd[p + "conv_after_body.weight"], sd[p + "conv_after_body.bias"], padding=1) + xf
    out = x + F.conv2d(res, sd[p + "conv_last.weight"], sd[p + "conv_last.bias"], padding=1)
    return out / r


# --------------------------------------------------------------------------------------------
# K10-K12  SearchTransfer / SelfTransfer  (model/SearchTransfer.py)
# --------------------------------------------------------------------------------------------
def correlation_max(lrsr: torch.Tensor, refsr: torch.Tensor, chunk: int = 4096):
    """``max_j <unfold3(ref)[j]/|.|, unfold3(lr)[i]/|.|>`` and its argmax (model/SearchTransfer.py:26-34).

    Same arithmetic as the reference's ``bmm`` + ``torch.max(dim=1)`` but evaluated in column chunks of the
    query index i so R (13 GB at 720p) is never resident; ties resolve to the lowest j like torch.max on CPU.
    """
    lu = F.normalize(F.unfold(lrsr, kernel_size=(3, 3), padding=1), dim=1)                      # [B,1152,N]
    ru = F.normalize(F.unfold(refsr, kernel_size=(3, 3), padding=1).permute(0, 2, 1), dim=2)    # [B,Nr,1152]
    n = lu.shape[2]
    ms, args = [], []
    for i0 in range(0, n, chunk):
        r = torch.bmm(ru, lu[:, :, i0:i0 + chunk])
        m, a = torch.max(r, dim=1)
        ms.append(m)
        args.append(a)
    return torch.cat(ms, dim=1), torch.cat(args, dim=1)      # (concatenation keeps dtype and the autograd graph of the maxima)


def _bis(inp: torch.Tensor, dim: int, index: torch.Tensor) -> torch.Tensor:
    """model/SearchTransfer.py:12-22."""
    views = [inp.size(0)] + [1 if i != dim else -1 for i in range(1, inp.dim())]
    expanse = list(inp.size())
    expanse[0] = -1
    expanse[dim] = -1
    return torch.gather(inp, dim, index.view(views).expand(expanse))


def search_transfer(lrsr3, refsr3, ref1, ref2, ref3, return_arg: bool = False):
    """``SearchTransfer.forward`` (model/SearchTransfer.py:24-51)."""
    smax, sarg = correlation_max(lrsr3, refsr3)
    h3, w3 = lrsr3.shape[-2:]
    t3 = F.fold(_bis(F.unfold(ref3, (3, 3), padding=1), 2, sarg), (h3, w3), (3, 3), padding=1) / 9.0
    t2 = F.fold(_bis(F.unfold(ref2, (6, 6), padding=2, stride=2), 2, sarg), (h3 * 2, w3 * 2), (6, 6), padding=2, stride=2) / 9.0
    t1 = F.fold(_bis(F.unfold(ref1, (12, 12), padding=4, stride=4), 2, sarg), (h3 * 4, w3 * 4), (12, 12), padding=4, stride=4) / 9.0
    s = smax.view(smax.size(0), 1, h3, w3)
    if return_arg:
        return s, t3, t2, t1, sarg
    return s, t3, t2, t1


def self_transfer(lrsr3, sd: SD, p: str = "SelfTransfer."):
    """``SelfTransfer.forward`` (model/SearchTransfer.py:59-79): reference = 90-degree rotated self."""
    refsr = lrsr3.transpose(2, 3).flip(2)
    smax, _ = correlation_max(lrsr3, refsr)
    s = smax.view(smax.size(0), 1, lrsr3.size(2), lrsr3.size(3))
    t3 = lrsr3
    t2 = F.relu(F.conv2d(F.interpolate(lrsr3, scale_factor=2, mode="bicubic"), sd[p + "search1.weight"], sd[p + "search1.bias"]))
    t1 = F.relu(F.conv2d(F.interpolate(t2, scale_factor=2, mode="bicubic"), sd[p + "search2.weight"], sd[p + "search2.bias"]))
    return s, t3, t2, t1


# --------------------------------------------------------------------------------------------
# decode and the two forward branches  (model/speinet.py)
# --------------------------------------------------------------------------------------------
def _c(x, sd, name, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding)


def decode(f_fusion, s, t3, t2, t1, sd: SD, cfg: Cfg):
    """``SPEINet._decode`` (model/speinet.py:92-120)."""
    up = lambda t, k: F.interpolate(t, scale_factor=k, mode="bicubic")
    f_lv3 = f_fusion + _c(torch.cat((f_fusion, t3), 1), sd, "conv_lv3") * s
    dec2 = _dec_stage(f_lv3, sd, "recons_net.decoder_second.", cfg.n_resblock)
    f_lv2 = dec2 + _c(torch.cat((dec2, t2), 1), sd, "conv_lv2") * up(s, 2)
    s1 = F.relu(_c(up(f_lv3, 2), sd, "search1"))
    s2 = F.relu(_c(f_lv2, sd, "search3", 1))
    s11 = F.relu(_c(torch.cat((dec2, s1), 1), sd, "search2"))
    s22 = F.relu(_c(torch.cat((f_lv2, s2), 1), sd, "search2"))
    f_v3 = dec2 + s11
    f_lv2 = f_lv2 + s22
    dec1 = _dec_stage(f_lv2, sd, "recons_net.decoder_first.", cfg.n_resblock)
    f_lv1 = dec1 + _c(torch.cat((dec1, t1), 1), sd, "conv_lv1") * up(s, 4)
    s13 = F.relu(_c(up(f_v3, 2), sd, "search13"))
    s23 = F.relu(_c(up(f_lv2, 2), sd, "search33", 1))
    s33 = F.relu(_c(f_lv1, sd, "search43", 1))
    f_lv1 = f_lv1 + F.relu(_c(torch.cat((s13, s23), 1), sd, "search33", 1)) \
        + F.relu(_c(torch.cat((s13, s33), 1), sd, "search33", 1)) \
        + F.relu(_c(torch.cat((s23, s33), 1), sd, "search33", 1))
    return out_block(f_lv1, sd, cfg)


def fused_features(x: torch.Tensor, sd: SD, cfg: Cfg) -> torch.Tensor:
    """f_mid, ``_process`` and the 1x1 ``fusion`` (model/speinet.py:75-90,129-134 == :141-146)."""
    n = cfg.n_sequence
    mid = x[:, n // 2]
    f_mid = enc(mid, sd, cfg) + enc(rl_prior(mid, 5, 0.01), sd, cfg)
    f_fusion = f_mid
    for i in range(n):
        if i == n // 2:
            continue
        feat = enc(x[:, i], sd, cfg) + enc(rl_prior(x[:, i], 1, 0.01), sd, cfg)
        f_fusion = torch.cat((f_fusion, swin(f_mid, feat, sd, cfg)), dim=1)
    return _c(f_fusion, sd, "fusion")


def forward_bs(x: torch.Tensor, sd: SD, cfg: Cfg) -> torch.Tensor:
    """``_forwardbs`` (model/speinet.py:122-136): reference features from frame n_sequence+1 (= x[:,4])."""
    sharp = x[:, cfg.n_sequence + 1]
    lv1 = in_block(sharp, sd, cfg)
    lv2 = encoder_first(lv1, sd, cfg)
    lv3 = encoder_second(lv2, sd, cfg)
    f_fusion = fused_features(x, sd, cfg)
    s, t3, t2, t1 = search_transfer(f_fusion, lv3, lv1, lv2, lv3)
    return decode(f_fusion, s, t3, t2, t1, sd, cfg)


def forward_b(x: torch.Tensor, sd: SD, cfg: Cfg) -> torch.Tensor:
    """``_forwardb`` (model/speinet.py:138-148): no sharp reference -> SelfTransfer."""
    f_fusion = fused_features(x, sd, cfg)
    s, t3, t2, t1 = self_transfer(f_fusion, sd)
    return decode(f_fusion, s, t3, t2, t1, sd, cfg)


def route(x: torch.Tensor) -> torch.Tensor:
    """``_forwardx`` (model/speinet.py:70-73): per sample, is frame 3 identically zero?"""
    return (x[:, 3] == 0).flatten(1).all(dim=1)


def forward(x: torch.Tensor, sd: SD, cfg: Cfg = Cfg()) -> torch.Tensor:
    """``SPEINet.forward`` (model/speinet.py:150-168).  x [B, n_sequence+2, 3, H, W] -> [B,3,H,W], unclamped."""
    z = route(x)
    out = torch.empty(x.shape[0], x.shape[2], x.shape[3], x.shape[4], dtype=x.dtype)
    if z.any():
        out[z] = forward_b(x[z], sd, cfg)
    if (~z).any():
        out[~z] = forward_bs(x[~z], sd, cfg)
    return out


def forward_swint(x: torch.Tensor, sd: SD, cfg: Cfg = Cfg()) -> torch.Tensor:
    """The `swint` variant, ``model/swint.py:51-67``: encoders without the RL prior, swin(f_mid, f_i) per neighbour frame,
    the 1x1 ``conv`` over the concatenation (n_sequence == 1: f_mid + swin(f_mid, f_mid)), then the plain decoder
    outBlock(decoder_first(decoder_second(.))).  x [B, >= n_sequence, 3, H, W] -> [B, 3, H, W]."""
    n = cfg.n_sequence
    f_mid = enc(x[:, n // 2], sd, cfg)
    f_fusion = f_mid
    for i in range(n):
        if i == n // 2:
            continue
        f_fusion = torch.cat((f_fusion, swin(f_mid, enc(x[:, i], sd, cfg), sd, cfg)), dim=1)
    if n == 1:
        f_fusion = f_fusion + swin(f_mid, f_mid, sd, cfg)
    f = _c(f_fusion, sd, "conv")
    f = _dec_stage(f, sd, "recons_net.decoder_second.", cfg.n_resblock)
    f = _dec_stage(f, sd, "recons_net.decoder_first.", cfg.n_resblock)
    return out_block(f, sd, cfg)


# --------------------------------------------------------------------------------------------
# metric helpers used by the harness (inference_SPEINet.py:477-500)
# --------------------------------------------------------------------------------------------
def to_uint8(t: torch.Tensor) -> torch.Tensor:
    """``tensor2numpy`` (inference_SPEINet.py:477-482): x255, clamp, round, uint8; CHW->HWC."""
    return (t[0] * 255.0).clamp(0, 255).round().to(torch.uint8).permute(1, 2, 0)


def psnr_uint8(a: torch.Tensor, b: torch.Tensor, shave: int = 4) -> float:
    """``calc_PSNR`` on 4-px-cropped uint8 HWC frames (inference_SPEINet.py:484-500)."""
    a = a[shave:-shave, shave:-shave].double()
    b = b[shave:-shave, shave:-shave].double()
    mse = torch.mean((a - b) ** 2).item()
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))
