"""Training step on HIP kernels: BASELINE.json config 5 (`trainer/trainer_swint.py` around `model/swint.py`)
and the full model (`trainer/trainer_swint_hsa_nsf.py` around `model/speinet.py`): the differentiable forward of
`speinet_amd.swint.SPEINet` / `speinet_amd.speinet.SPEINet` in train() mode, built from `torch.autograd.Function`s whose forward
AND backward are the C-ABI kernels, on the parameters of the drop-in module itself — so the reference trainer's own four lines
work unchanged (trainer/trainer_swint.py:39-44, trainer_swint_hsa_nsf.py:31-40):

    out = model(input); optimizer.zero_grad(); loss = loss_fn(out, gt); loss.backward(); optimizer.step()

What is train-mode about the graph (all three differ from the eval() graph the inference path runs):
  * BatchNorm2d(1) of the ResBlock gates (model/block.py:49-68) normalises with the statistics of the batch and updates its
    running buffers (momentum 0.01, unbiased variance, num_batches_tracked);
  * DropPath (model/swinir.py:203,278-279; `timm.models.layers.DropPath`, a dependency the reference does not pin — its
    published algorithm: one Bernoulli(keep) draw per sample and call, divided by keep) scales each Swin branch per sample.
    `drop_path_scales` draws the factors from torch's CPU generator in the reference's call order; tests pass the reference's
    own draws in explicitly;
  * nothing else: Dropout / attention dropout have p = 0 in the reference's SwinIR constructor call (model/swint.py:31-40).

Activations are NHWC pixel rows, the batch folded into the row index: [B * H * W, C] fp32.  Convolutions and window attention
loop over the samples (their kernels take one map); linears, LayerNorm, GELU take all rows at once.

What runs where:
  HIP    every pixel-sized operation: conv / transposed conv / linear forward and data gradient (spei_igemm_f32, spei_conv5_in),
         weight + bias gradients (spei_conv_wgrad_f32), ReLU / GELU masks, LayerNorm forward and backward, window attention
         forward and backward, the gates' plane statistics, the gated residual sum and its backward, DropPath row scaling,
         the Richardson-Lucy prior, SearchTransfer forward (exact f32 correlation arg-max, gathers) and backward, bicubic
         up-sampling and its adjoint, the `* weight_S` row scale and its gradient
  torch  parameter-sized or [H][C]-plane-sized work: the gate maps (SE MLP, two 2->1 convs, BatchNorm(1)) and their autograd,
         the relative-position bias gather, weight re-layouts, sums over per-block / per-window partial gradients, the
         ConvTranspose2d bias gradient (a column sum), concatenations (`torch.cat`) and residual additions of feature rows, the
         rotated view SelfTransfer searches, the per-position query lists of the search backward (a stable sort), the routing
         scatter of the two sample classes, the loss on the [B,3,H,W] output, the optimizer (the reference's own torch.optim.Adam)
No CPU path: everything here raises without the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib
from .ops import ACT_NONE, ACT_RELU, CONV, CONV_T, Ctx

_NULL = C.c_void_p(0)


def _ctx(device) -> Ctx:
    return Ctx("f32", "bf16x3", device=device)


# Arithmetic of the step's forward and data-gradient GEMMs (`model.train_precision`): "f32" — v_mfma_f32_32x32x2_f32, the form
# G20 - G22 pin to 1e-6; "bf16x3" — split products on the 16-bit matrix pipe (csrc/conv_slab16.hip, the inference kernels' SPLIT form:
# 16-bit-significand operands, 2^-16 relative per product, fp32 accumulation): several times the rate, and still finer than the
# arithmetic the reference trains in (main_SPEINet.py:12 `set_float32_matmul_precision('medium')`: TF32 / bf16 matmuls).  Weight
# gradients, the stride-2 transposed forms, attention and everything pixel-sized stay fp32.  The choice is read when the graph is
# built (a context variable around forward_swint / forward_speinet) and kept by every Function for its backward: the autograd
# engine runs those on its own thread.
import contextvars

_PREC = contextvars.ContextVar("speinet_train_precision", default="f32")


def _split_frags(ctx: "Ctx", owner):
    """(hi, lo) bf16 halves of a GEMM weight in MFMA fragment order, straight from the parameter in the reference's layout (ONE launch,
    spei_pack_split16).  `owner` = (parameter tensor, tag): tag "fwd" = the forward weight [tap][N][K], "dgrad" = the stride-1
    data-gradient weight (taps reversed, channel axes swapped).  The pair is kept ON the parameter object together with the `_version`
    it was made from — the same weight is packed once per optimizer step however many encoder passes use it, and nothing can outlive
    or be mistaken for another parameter (an address-keyed table can: a freed parameter's address is handed to the next model's).
    The stamp is (`_version`, storage address, device): `.to()` / `param.data = t` swap the storage without touching the counter.
    Edits THROUGH `.data` (`p.data.mul_()`) change neither: the model's `invalidate_packed()` — run at the top of every autograd
    forward — drops the cache of every parameter (`drop_split_cache`), so a pair lives for one forward + backward at most."""
    weight, tag = owner
    stamp = (weight._version, weight.data_ptr(), str(weight.device))
    cache = getattr(weight, "_spei_split", None)
    if cache is not None:
        hit = cache.get(tag)
        if hit is not None and hit[0] == stamp:
            return hit[1]
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    assert w.dtype == torch.float32 and w.dim() in (2, 4) and (w.dim() == 2 or w.shape[2] == w.shape[3])
    n, k, ks = w.shape[0], w.shape[1], (w.shape[2] if w.dim() == 4 else 1)
    hi = torch.empty(w.numel(), device=w.device, dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    _lib.check(_lib.lib().spei_pack_split16(ctx._tp(w), n, k, ks, {"fwd": 0, "dgrad": 1}[tag], ctx._tp(hi), ctx._tp(lo), ctx._stream()),
               "spei_pack_split16")
    out = (hi, lo)
    try:
        if cache is None:
            cache = weight._spei_split = {}
        cache[tag] = (stamp, out)
    except AttributeError:
        pass
    return out


def drop_split_cache(module) -> None:
    """Forget the packed bf16 halves kept on the module's parameters (called by `invalidate_packed()`)."""
    for p_ in module.parameters():
        if getattr(p_, "_spei_split", None):
            p_._spei_split = {}


def _wkey(weight: torch.Tensor, tag: str):
    return (weight, tag)


def _p(ctx: Ctx, t: Optional[torch.Tensor]):
    if t is None:
        return _NULL
    assert t.dtype == torch.float32
    return ctx._tp(t)


# ---- the GEMM family on raw tensors (no packing objects: the weights change every step) ---------------------------------------
def _w_conv(weight: torch.Tensor) -> torch.Tensor:
    """Conv2d [N, K, kh, kw] -> [tap][N][K]."""
    n, k, kh, kw = weight.shape
    return weight.detach().permute(2, 3, 0, 1).reshape(kh * kw, n, k).contiguous()


def _igemm(ctx: Ctx, a: torch.Tensor, K: int, w_tnk: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, N: int,
           Hin: int, Win: int, Hout: int, Wout: int, ksize: int, stride: int, mode: int, act: int,
           residual: Optional[torch.Tensor] = None, rowscale: Optional[torch.Tensor] = None, batch: int = 1, prec: str = "f32",
           wkey=None) -> None:
    """One launch over `batch` equally sized maps stored one after the other (blockIdx.z / .y = sample).  `w_tnk`: the [tap][N][K] fp32
    weight or a callable that makes it (only the fp32 kernel needs it: the split path packs from the parameter behind `wkey`)."""
    assert a.is_contiguous() and out.is_contiguous()
    assert a.shape == (batch * Hin * Win, K) and out.shape == (batch * Hout * Wout, N) and K % 32 == 0 and N % 32 == 0
    assert residual is None or (residual.is_contiguous() and residual.shape == out.shape)
    lib = _lib.lib()
    if prec == "bf16x3" and mode == CONV and (rowscale is None or batch == 1):
        from .pack import BF16, F32
        fhi, flo = _split_frags(ctx, wkey)
        ldr = N if residual is not None else 0
        if batch == 1 or Wout == 1:
            rows = batch * Hin * Win
            dims = (rows, 1, batch * Hout * Wout, 1) if (ksize == 1 and stride == 1) else (Hin, Win, Hout, Wout)
            assert batch == 1 or (ksize == 1 and stride == 1)
            _lib.check(lib.spei_conv_slab16(BF16, _p(ctx, a), K, K, _NULL, 0, 0, F32, ctx._tp(fhi), ctx._tp(flo), _p(ctx, bias), _p(ctx, out), N, F32,
                                            _p(ctx, residual), ldr, _p(ctx, rowscale), *dims, N, ksize, stride, ksize // 2, act, 0, ctx._stream()),
                       "spei_conv_slab16")
        else:
            _lib.check(lib.spei_conv_slab16_batched(BF16, _p(ctx, a), K, F32, ctx._tp(fhi), ctx._tp(flo), _p(ctx, bias), _p(ctx, out), F32,
                                                    _p(ctx, residual), batch, Hin, Win, Hout, Wout, N, ksize, stride, ksize // 2, act,
                                                    ctx._stream()), "spei_conv_slab16_batched")
        return
    w_tnk = w_tnk() if callable(w_tnk) else w_tnk
    assert w_tnk.is_contiguous() and tuple(w_tnk.shape) == (ksize * ksize, N, K), (tuple(w_tnk.shape), ksize, N, K)
    _lib.check(lib.spei_igemm_f32_batched(_p(ctx, a), K, K, _NULL, 0, 0, _p(ctx, w_tnk), _p(ctx, bias), _p(ctx, out), N,
                                          _p(ctx, residual), N if residual is not None else 0, _p(ctx, rowscale), Hin, Win, Hout, Wout, N,
                                          ksize, stride, ksize // 2, mode, act, batch, ctx._stream()), "spei_igemm_f32_batched")


def _wgrad(ctx: Ctx, x: torch.Tensor, K: int, dy: torch.Tensor, N: int, Hin: int, Win: int, Hout: int, Wout: int, ksize: int,
           stride: int, want_bias: bool = True, batch: int = 1, prec: str = "f32"):
    """dw [tap][N][K], db [N] of a Conv2d(K -> N) from its input rows x and output-gradient rows dy, summed over `batch` maps."""
    lib = _lib.lib()
    dev = x.device
    assert x.shape == (batch * Hin * Win, K) and dy.shape == (batch * Hout * Wout, N)
    dw = torch.empty(ksize * ksize, N, K, device=dev)
    db = torch.empty(N, device=dev) if want_bias else None
    for n0 in range(0, N, 256):                                       # the kernel takes at most 256 output channels per call
        nn_ = min(256, N - n0)
        dwp = dw if N <= 256 else torch.empty(ksize * ksize, nn_, K, device=dev)
        dbp = (db if N <= 256 else torch.empty(nn_, device=dev)) if want_bias else None
        ws = torch.empty(lib.spei_wgrad_ws_floats(Hout, Wout, nn_, K, ksize), device=dev)
        dyp = C.c_void_p(dy.data_ptr() + 4 * n0)
        assert dy.device == ctx.device and dy.is_contiguous() and dy.dtype == torch.float32
        fn = lib.spei_conv_wgrad_bf16x3_batched if prec == "bf16x3" else lib.spei_conv_wgrad_f32_batched
        _lib.check(fn(_p(ctx, x), K, dyp, N, _p(ctx, dwp), _p(ctx, dbp), _p(ctx, ws), Hin, Win, Hout, Wout, nn_,
                      K, ksize, stride, ksize // 2, batch, ctx._stream()), "spei_conv_wgrad_batched")
        if N > 256:
            dw[:, n0:n0 + nn_] = dwp
            if want_bias:
                db[n0:n0 + nn_] = dbp
    return dw, db


def _relu_mask(ctx: Ctx, y: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    dz = torch.empty_like(dy)
    _lib.check(_lib.lib().spei_relu_bwd(_p(ctx, y), _p(ctx, dy), _p(ctx, dz), dy.numel(), ctx._stream()), "spei_relu_bwd")
    return dz


class _Conv2d(torch.autograd.Function):
    """y = act(conv(x, weight) + bias) [+ residual] on a batch of maps: x [B*H*W, K] -> [B*Ho*Wo, N]; Conv2d(K -> N, k, stride,
    padding k // 2) (model/recons_video_ori.py:28-71, model/block.py:26-47, model/swinir.py:467,667,716,742)."""

    @staticmethod
    def forward(fctx, x, weight, bias, residual, B, H, W, ksize, stride, relu):
        ctx = _ctx(x.device)
        n, k = weight.shape[0], weight.shape[1]
        ho, wo = (H + 2 * (ksize // 2) - ksize) // stride + 1, (W + 2 * (ksize // 2) - ksize) // stride + 1
        x = x.contiguous()
        b = bias.detach().contiguous()
        out = torch.empty(B * ho * wo, n, device=x.device)
        res = residual.contiguous() if residual is not None else None
        prec = _PREC.get()
        _igemm(ctx, x, k, lambda: _w_conv(weight), b, out, n, H, W, ho, wo, ksize, stride, CONV, ACT_RELU if relu else ACT_NONE, res, batch=B,
               prec=prec, wkey=_wkey(weight, "fwd"))
        fctx.save_for_backward(x, weight, out if relu else None)
        fctx.meta = (B, H, W, ho, wo, ksize, stride, relu, residual is not None)
        fctx.prec = prec
        return out

    @staticmethod
    def backward(fctx, dy):
        x, weight, y = fctx.saved_tensors
        B, H, W, ho, wo, ksize, stride, relu, has_res = fctx.meta
        ctx = _ctx(dy.device)
        n, k = weight.shape[0], weight.shape[1]
        dy = dy.contiguous()
        dres = dy if has_res else None
        assert not (relu and has_res)
        dz = _relu_mask(ctx, y, dy) if relu else dy
        dw, db = _wgrad(ctx, x, k, dz, n, H, W, ho, wo, ksize, stride, batch=B, prec=fctx.prec)
        dweight = dw.view(ksize, ksize, n, k).permute(2, 3, 0, 1).contiguous()
        dx = None
        if fctx.needs_input_grad[0]:
            # data gradient = transposed convolution of dZ with the weights' channel axes swapped (include/speinet_hip.h)
            wt = lambda: _w_conv(weight).transpose(1, 2).contiguous()                      # [t][k][n]
            hf, wf = ho * stride, wo * stride
            dx = torch.empty(B * hf * wf, k, device=dy.device)
            if fctx.prec == "bf16x3" and stride == 1:
                # stride 1: the transposed convolution IS the convolution with the taps reversed
                _igemm(ctx, dz, n, lambda: wt().flip(0).contiguous(), None, dx, k, ho, wo, hf, wf, ksize, 1, CONV, ACT_NONE, batch=B,
                       prec="bf16x3", wkey=_wkey(weight, "dgrad"))
            else:
                _igemm(ctx, dz, n, wt, None, dx, k, ho, wo, hf, wf, ksize, stride, CONV_T, ACT_NONE, batch=B)
            if (hf, wf) != (H, W):        # odd input size under stride 2: the transposed conv made one row / column too many
                dx = dx.view(B, hf, wf, k)[:, :H, :W].reshape(B * H * W, k).contiguous()
        return dx, dweight, db, dres, None, None, None, None, None, None


class _ConvT2d(torch.autograd.Function):
    """y = relu(ConvTranspose2d(K -> N, 3, stride 2, padding 1, output_padding 1)(x) + bias)   (model/recons_video_ori.py:58-71,
    the decoder tails): x [B*H*W, K] -> [B*2H*2W, N].  Its data gradient is the stride-2 Conv2d of dY with the SAME weight tensor
    read as [out = K][in = N][3][3]; its weight gradient is that convolution's weight gradient with dY as the input map."""

    @staticmethod
    def forward(fctx, x, weight, bias, B, H, W):
        ctx = _ctx(x.device)
        k, n = weight.shape[0], weight.shape[1]
        x = x.contiguous()
        w = weight.detach().permute(2, 3, 1, 0).reshape(9, n, k).contiguous()               # [tap][N][K]
        out = torch.empty(B * 4 * H * W, n, device=x.device)
        b = bias.detach().contiguous()
        _igemm(ctx, x, k, w, b, out, n, H, W, 2 * H, 2 * W, 3, 2, CONV_T, ACT_RELU, batch=B)
        fctx.save_for_backward(x, weight, out)
        fctx.meta = (B, H, W)
        return out

    @staticmethod
    def backward(fctx, dy):
        x, weight, y = fctx.saved_tensors
        B, H, W = fctx.meta
        ctx = _ctx(dy.device)
        k, n = weight.shape[0], weight.shape[1]
        dz = _relu_mask(ctx, y, dy.contiguous())
        wc = _w_conv(weight)                                                               # Conv2d view: [tap][out = K][in = N]
        dx = torch.empty(B * H * W, k, device=dy.device)
        _igemm(ctx, dz, n, wc, None, dx, k, 2 * H, 2 * W, H, W, 3, 2, CONV, ACT_NONE, batch=B)
        dw, _ = _wgrad(ctx, dz, n, x, k, 2 * H, 2 * W, H, W, 3, 2, want_bias=False, batch=B)                            # [t][K][N]
        dweight = dw.view(3, 3, k, n).permute(2, 3, 0, 1).contiguous()
        return dx, dweight, dz.sum(dim=0), None, None, None


class _ConvIn(torch.autograd.Function):
    """The 3-channel head conv + ReLU on NCHW frames (spei_conv5_in): frames [B,3,H,W] -> [B*H*W, 32]."""

    @staticmethod
    def forward(fctx, frames, weight, bias):
        ctx = _ctx(frames.device)
        B, c, H, W = frames.shape
        n = weight.shape[0]
        frames = frames.contiguous()
        w = _w_conv(weight)
        b = bias.detach().contiguous()
        out = torch.empty(B * H * W, n, device=frames.device)
        lib = _lib.lib()
        for i in range(B):
            _lib.check(lib.spei_conv5_in(_p(ctx, frames[i]), _p(ctx, w), _p(ctx, b), _p(ctx, out[i * H * W:(i + 1) * H * W]), H, W, n,
                                         ctx._stream()), "spei_conv5_in")
        fctx.save_for_backward(frames.permute(0, 2, 3, 1).reshape(B * H * W, c).contiguous(), weight, out)
        fctx.meta = (B, H, W)
        return out

    @staticmethod
    def backward(fctx, dy):
        x, weight, y = fctx.saved_tensors
        B, H, W = fctx.meta
        ctx = _ctx(dy.device)
        n, k = weight.shape[0], weight.shape[1]
        dz = _relu_mask(ctx, y, dy.contiguous())
        dw, db = _wgrad(ctx, x, k, dz, n, H, W, H, W, 5, 1, batch=B)
        return None, dw.view(5, 5, n, k).permute(2, 3, 0, 1).contiguous(), db


class _Linear(torch.autograd.Function):
    """out = residual + rowscale * (x W^T + b)   (nn.Linear; residual and the per-row DropPath factor optional, both in the GEMM's
    epilogue).  x [M, K], weight [N, K]."""

    @staticmethod
    def forward(fctx, x, weight, bias, residual, rowscale):
        ctx = _ctx(x.device)
        n, k = weight.shape
        x = x.contiguous()
        M = x.shape[0]
        out = torch.empty(M, n, device=x.device)
        prec = _PREC.get()
        _igemm(ctx, x, k, lambda: weight.detach().reshape(1, n, k).contiguous(), bias.detach().contiguous(), out, n, M, 1, M, 1, 1, 1, CONV, ACT_NONE,
               residual.contiguous() if residual is not None else None, rowscale, prec=prec, wkey=_wkey(weight, "fwd"))
        fctx.save_for_backward(x, weight, rowscale)
        fctx.has_res = residual is not None
        fctx.prec = prec
        return out

    @staticmethod
    def backward(fctx, dy):
        x, weight, rowscale = fctx.saved_tensors
        ctx = _ctx(dy.device)
        n, k = weight.shape
        M = x.shape[0]
        dy = dy.contiguous()
        g = dy
        if rowscale is not None:
            g = torch.empty_like(dy)
            _lib.check(_lib.lib().spei_scale_rows(_p(ctx, dy), _p(ctx, rowscale), _p(ctx, g), M, n, ctx._stream()), "spei_scale_rows")
        dw, db = _wgrad(ctx, x, k, g, n, 1, M, 1, M, 1, 1, prec=fctx.prec)       # a 1x1 layer: the token list as ONE row of M pixels
        dx = None
        if fctx.needs_input_grad[0]:
            dx = torch.empty(M, k, device=dy.device)
            _igemm(ctx, g, n, lambda: weight.detach().t().reshape(1, k, n).contiguous(), None, dx, k, M, 1, M, 1, 1, 1, CONV, ACT_NONE, prec=fctx.prec,
                   wkey=_wkey(weight, "dgrad"))
        return dx, dw.view(n, k), db, (dy if fctx.has_res else None), None


class _LayerNorm(torch.autograd.Function):
    """nn.LayerNorm(256) over the channels of [M, 256] rows."""

    @staticmethod
    def forward(fctx, x, gamma, beta):
        ctx = _ctx(x.device)
        x = x.contiguous()
        assert x.shape[1] == 256
        y = torch.empty_like(x)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        _lib.check(_lib.lib().spei_layernorm256(_p(ctx, x), _p(ctx, y), 0, _p(ctx, g), _p(ctx, b), x.shape[0], ctx._stream()), "spei_layernorm256")
        fctx.save_for_backward(x, gamma)
        return y

    @staticmethod
    def backward(fctx, dy):
        x, gamma = fctx.saved_tensors
        ctx = _ctx(dy.device)
        lib = _lib.lib()
        M = x.shape[0]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        part = torch.empty(lib.spei_ln_bwd_blocks(M), 2, 256, device=dy.device)
        _lib.check(lib.spei_layernorm256_bwd(_p(ctx, x), _p(ctx, gamma.detach().contiguous()), _p(ctx, dy), _p(ctx, dx), _p(ctx, part), M,
                                             ctx._stream()), "spei_layernorm256_bwd")
        s = part.sum(dim=0)
        return dx, s[0], s[1]


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(fctx, pre):
        ctx = _ctx(pre.device)
        pre = pre.contiguous()
        out = torch.empty_like(pre)
        _lib.check(_lib.lib().spei_gelu_fwd(_p(ctx, pre), _p(ctx, out), pre.numel(), ctx._stream()), "spei_gelu_fwd")
        fctx.save_for_backward(pre)
        return out

    @staticmethod
    def backward(fctx, dy):
        (pre,) = fctx.saved_tensors
        ctx = _ctx(dy.device)
        d = torch.empty_like(pre)
        _lib.check(_lib.lib().spei_gelu_bwd(_p(ctx, pre), _p(ctx, dy.contiguous()), _p(ctx, d), pre.numel(), ctx._stream()), "spei_gelu_bwd")
        return d


class _WindowAttention(torch.autograd.Function):
    """softmax(q k^T + bias + mask) v per 5x5 window and head (model/swinir.py:115-149); q [B*H*W, 256] already scaled by
    head_dim^-0.5, kv [B*H*W, 512], relbias [8, 25, 25]."""

    @staticmethod
    def forward(fctx, q, kv, relbias, B, H, W, shift):
        ctx = _ctx(q.device)
        q, kv, rb = q.contiguous(), kv.contiguous(), relbias.detach().contiguous()
        out = torch.empty_like(q)
        lib = _lib.lib()
        hw = H * W
        _lib.check(lib.spei_window_attention_batched(_p(ctx, q), _p(ctx, kv), 0, _p(ctx, rb), _p(ctx, out), H, W, shift, B, ctx._stream()),
                   "spei_window_attention_batched")
        fctx.save_for_backward(q, kv, rb)
        fctx.meta = (B, H, W, shift)
        return out

    @staticmethod
    def backward(fctx, dout):
        q, kv, rb = fctx.saved_tensors
        B, H, W, shift = fctx.meta
        ctx = _ctx(dout.device)
        lib = _lib.lib()
        dout = dout.contiguous()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        nwin = (H // 5) * (W // 5)
        part = torch.empty(B, nwin, 8, 25, 25, device=dout.device)
        hw = H * W
        _lib.check(lib.spei_window_attention_bwd(_p(ctx, q), _p(ctx, kv), _p(ctx, rb), _p(ctx, dout), _p(ctx, dq), _p(ctx, dkv), _p(ctx, part),
                                                 H, W, shift, B, ctx._stream()), "spei_window_attention_bwd")
        return dq, dkv, part.sum(dim=(0, 1)), None, None, None, None


class _RowScale(torch.autograd.Function):
    """y[m][n] = x[m][n] * s[m]   (`* weight_S` and its bicubic up-samplings, model/speinet.py:93,95,104); both inputs differentiable."""

    @staticmethod
    def forward(fctx, x, s):
        ctx = _ctx(x.device)
        x, s = x.contiguous(), s.contiguous()
        M, n = x.shape
        assert s.numel() == M
        out = torch.empty_like(x)
        _lib.check(_lib.lib().spei_scale_rows(_p(ctx, x), _p(ctx, s), _p(ctx, out), M, n, ctx._stream()), "spei_scale_rows")
        fctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(fctx, dy):
        x, s = fctx.saved_tensors
        ctx = _ctx(dy.device)
        lib = _lib.lib()
        M, n = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.spei_scale_rows(_p(ctx, dy), _p(ctx, s), _p(ctx, dx), M, n, ctx._stream()), "spei_scale_rows")
        ds = torch.empty_like(s)
        _lib.check(lib.spei_rowdot(_p(ctx, dy), _p(ctx, x), _p(ctx, ds), M, n, ctx._stream()), "spei_rowdot")
        return dx, ds


class _Bicubic(torch.autograd.Function):
    """F.interpolate(scale_factor=s, mode='bicubic') on pixel rows: [B*H*W, C] -> [B*sH*sW, C]   (model/speinet.py:95-113)."""

    @staticmethod
    def forward(fctx, x, B, H, W, s):
        ctx = _ctx(x.device)
        x = x.contiguous()
        c = x.shape[1]
        out = torch.empty(B * s * H * s * W, c, device=x.device)
        lib = _lib.lib()
        for i in range(B):
            _lib.check(lib.spei_upsample_bicubic(_p(ctx, x[i * H * W:(i + 1) * H * W]), c, _p(ctx, out[i * s * s * H * W:(i + 1) * s * s * H * W]), c,
                                                 H, W, c, s, ACT_NONE, ctx._stream()), "spei_upsample_bicubic")
        fctx.meta = (B, H, W, s, c)
        return out

    @staticmethod
    def backward(fctx, dy):
        B, H, W, s, c = fctx.meta
        ctx = _ctx(dy.device)
        dy = dy.contiguous()
        dx = torch.empty(B * H * W, c, device=dy.device)
        lib = _lib.lib()
        for i in range(B):
            _lib.check(lib.spei_upsample_bicubic_bwd(_p(ctx, dy[i * s * s * H * W:(i + 1) * s * s * H * W]), _p(ctx, dx[i * H * W:(i + 1) * H * W]),
                                                     H, W, c, s, ctx._stream()), "spei_upsample_bicubic_bwd")
        return dx, None, None, None, None


class _SearchTransfer(torch.autograd.Function):
    """S, T_lv3, T_lv2, T_lv1 of model/SearchTransfer.py:24-50 on pixel rows: lr [B*H*W, 128] queries; ref3 [B*Hr*Wr, 128] the map
    that is searched AND gathered at level 3, ref2 [B*2Hr*2Wr, 64], ref1 [B*4Hr*4Wr, 32] gathered at the other two levels (pass
    None for all three outputs T when only S is wanted: SelfTransfer, :58-79).  S [B*H*W]."""

    @staticmethod
    def forward(fctx, lr, ref3, ref2, ref1, B, H, W, Hr, Wr):
        ctx = _ctx(lr.device)
        lib = _lib.lib()
        dev = lr.device
        lr, ref3 = lr.contiguous(), ref3.contiguous()
        c = lr.shape[1]
        hw, hwr = H * W, Hr * Wr
        want_t = ref2 is not None
        inv_lr, inv_ref = torch.empty(B * hw, device=dev), torch.empty(B * hwr, device=dev)
        S = torch.empty(B * hw, device=dev)
        arg = torch.empty(B * hw, device=dev, dtype=torch.int32)
        ws = torch.empty(lib.spei_corr_ws_floats(hw), device=dev)
        T = [torch.empty(B * s * s * hw, cc, device=dev) for s, cc in ((1, c), (2, c // 2), (4, c // 4))] if want_t else [None] * 3
        refs = [ref3, ref2.contiguous(), ref1.contiguous()] if want_t else None
        for i in range(B):
            a, r = lr[i * hw:(i + 1) * hw], ref3[i * hwr:(i + 1) * hwr]
            il, ir = inv_lr[i * hw:(i + 1) * hw], inv_ref[i * hwr:(i + 1) * hwr]
            _lib.check(lib.spei_patch_invnorm(_p(ctx, a), c, _p(ctx, il), H, W, c, ctx._stream()), "spei_patch_invnorm")
            _lib.check(lib.spei_patch_invnorm(_p(ctx, r), c, _p(ctx, ir), Hr, Wr, c, ctx._stream()), "spei_patch_invnorm")
            _lib.check(lib.spei_corr_argmax(_p(ctx, a), c, _p(ctx, r), c, _p(ctx, il), _p(ctx, ir), H, W, Hr, Wr, c, _p(ctx, S[i * hw:(i + 1) * hw]),
                                            ctx._tp(arg[i * hw:(i + 1) * hw]), _p(ctx, ws), ctx._stream()), "spei_corr_argmax")
            if want_t:
                for (s, t, rf) in zip((1, 2, 4), T, refs):
                    cc = rf.shape[1]
                    _lib.check(lib.spei_gather_fold(_p(ctx, rf[i * s * s * hwr:(i + 1) * s * s * hwr]), cc, ctx._tp(arg[i * hw:(i + 1) * hw]),
                                                    _p(ctx, t[i * s * s * hw:(i + 1) * s * s * hw]), cc, H, W, Hr, Wr, cc, s, ctx._stream()),
                               "spei_gather_fold")
        fctx.save_for_backward(lr, ref3, inv_lr, inv_ref, S, arg)
        fctx.meta = (B, H, W, Hr, Wr, want_t, c)
        fctx.mark_non_differentiable(arg)
        return S, T[0], T[1], T[2], arg

    @staticmethod
    def backward(fctx, dS, dT3, dT2, dT1, _darg):
        lr, ref3, inv_lr, inv_ref, S, arg = fctx.saved_tensors
        B, H, W, Hr, Wr, want_t, c = fctx.meta
        ctx = _ctx(lr.device)
        lib = _lib.lib()
        dev = lr.device
        hw, hwr = H * W, Hr * Wr
        dS = dS.contiguous() if dS is not None else torch.zeros_like(S)
        dlr = torch.empty_like(lr)
        dref = [torch.empty(B * s * s * hwr, cc, device=dev) for s, cc in ((1, c), (2, c // 2), (4, c // 4))] if want_t else \
            [torch.empty(B * hwr, c, device=dev), None, None]
        dT = [t.contiguous() if t is not None else None for t in (dT3, dT2, dT1)]
        for i in range(B):
            sl, slr = slice(i * hw, (i + 1) * hw), slice(i * hwr, (i + 1) * hwr)
            a = arg[sl].long()
            # per reference position, the list of queries that chose it (stable sort: fixed summation order)
            order = torch.sort(a, stable=True)[1].to(torch.int32).contiguous()
            start = torch.zeros(hwr + 1, device=dev, dtype=torch.int64)
            start[1:] = torch.cumsum(torch.bincount(a, minlength=hwr), 0)
            start = start.to(torch.int32).contiguous()
            _lib.check(lib.spei_corr_s_bwd_lr(_p(ctx, lr[sl]), _p(ctx, ref3[slr]), _p(ctx, inv_lr[sl]), _p(ctx, inv_ref[slr]), _p(ctx, S[sl]),
                                              ctx._tp(arg[sl]), _p(ctx, dS[sl]), _p(ctx, dlr[sl]), H, W, Hr, Wr, c, ctx._stream()),
                       "spei_corr_s_bwd_lr")
            for j, s in enumerate((1, 2, 4)):
                if dref[j] is None:
                    continue
                cc = dref[j].shape[1]
                dt = dT[j][i * s * s * hw:(i + 1) * s * s * hw] if dT[j] is not None else None
                if s > 1 and dt is None:
                    dref[j][i * s * s * hwr:(i + 1) * s * s * hwr].zero_()
                    continue
                first = s == 1
                _lib.check(lib.spei_search_bwd_ref(_p(ctx, ref3[slr]) if first else _NULL, _p(ctx, dt), _p(ctx, lr[sl]) if first else _NULL,
                                                   _p(ctx, inv_lr[sl]) if first else _NULL, _p(ctx, inv_ref[slr]) if first else _NULL,
                                                   _p(ctx, S[sl]) if first else _NULL, _p(ctx, dS[sl]) if first else _NULL, ctx._tp(order),
                                                   ctx._tp(start), _p(ctx, dref[j][i * s * s * hwr:(i + 1) * s * s * hwr]), H, W, Hr, Wr, cc, s,
                                                   ctx._stream()), "spei_search_bwd_ref")
        return dlr, dref[0], dref[1], dref[2], None, None, None, None, None


# ---- the ResBlock's gated residual sum, batch form, BatchNorm(1) in either mode -------------------------------------------------
def _gate_ptrs(ctx: Ctx, params):
    """(prm, run) pointer arrays of spei_gate_maps_fwd / _bwd from the 14 gate parameters of a ResBlock, and the tensors they point to
    (kept alive by the caller): se_w1, se_b1, se_w2, se_b2, cw_w, cw_g, cw_b, cw_rm, cw_rv, hc_w, hc_g, hc_b, hc_rm, hc_rv."""
    t = [q.detach() for q in params]
    t = [q if q.is_contiguous() else q.contiguous() for q in t]
    assert all(q.dtype == torch.float32 and q.device == ctx.device for q in t)
    order = (0, 1, 2, 3, 4, 5, 6, 9, 10, 11)
    prm = (C.c_void_p * 10)(*[q.data_ptr() for q in (t[i] for i in order)])
    run = (C.c_void_p * 4)(*[t[i].data_ptr() for i in (7, 8, 12, 13)])
    return prm, run, t


class _GatedSum(torch.autograd.Function):
    """out = x + x1 * (s + g1[y] + g2[x]) per sample, the gates built from x1's plane statistics (the tail of model/block.py:136-140).
    x, x1 [B*H*W, C].  Statistics, gate maps (SE MLP, the two 2 -> 1 channel convolutions, BatchNorm2d(1) on batch statistics in
    train() mode with its running buffers moved) and the gated sum are HIP kernels forward and backward (csrc/resblock.hip,
    backward.hip, gates_train.hip); `groups`: BatchNorm statistics per group of B / groups consecutive samples (one group per encoder
    pass when several passes share a launch)."""

    @staticmethod
    def forward(fctx, x, x1, B, H, W, bn_train, groups, *params):
        ctx = _ctx(x.device)
        lib = _lib.lib()
        c = x.shape[1]
        dev = x.device
        x, x1 = x.contiguous(), x1.contiguous()
        rowmax, rowmean = torch.empty(B, H, c, device=dev), torch.empty(B, H, c, device=dev)
        colmax, colmean = torch.empty(B, W, c, device=dev), torch.empty(B, W, c, device=dev)
        mean = torch.empty(B, c, device=dev)
        ws = torch.empty(B * lib.spei_plane_ws_floats(H, W, c), device=dev)
        _lib.check(lib.spei_plane_stats_batched(_p(ctx, x1), _NULL, 0, H, W, c, _p(ctx, rowmax), _p(ctx, rowmean), _p(ctx, colmax),
                                                _p(ctx, colmean), _p(ctx, mean), _p(ctx, ws), B, ctx._stream()), "spei_plane_stats_batched")
        prm, run, keep_alive = _gate_ptrs(ctx, params)
        s, g1, g2 = torch.empty(B, c, device=dev), torch.empty(B, H, c, device=dev), torch.empty(B, W, c, device=dev)
        saved = torch.empty(lib.spei_gate_train_saved_floats(B, groups, H, W, c), device=dev)
        gws = torch.empty(lib.spei_gate_train_ws_floats(B, groups, H, W, c) // 2 + 1, device=dev, dtype=torch.float64)
        _lib.check(lib.spei_gate_maps_fwd(_p(ctx, rowmax), _p(ctx, rowmean), _p(ctx, colmax), _p(ctx, colmean), _p(ctx, mean), prm, run, B, groups,
                                          H, W, c, int(bn_train), int(bn_train), _p(ctx, s), _p(ctx, g1), _p(ctx, g2), _p(ctx, saved),
                                          C.c_void_p(gws.data_ptr()), ctx._stream()), "spei_gate_maps_fwd")
        out = torch.empty_like(x)
        _lib.check(lib.spei_resblock_apply_batched(_p(ctx, x), _p(ctx, x1), 0, _p(ctx, s), _p(ctx, g1), _p(ctx, g2), _p(ctx, out), B, H, W, c,
                                                   ctx._stream()), "spei_resblock_apply_batched")
        # the backward reads the SAVED batch statistics (train) / the running statistics as they were (eval: copied into `saved` by the
        # forward), never the running buffers, which move again when the same block runs on the next frame
        fctx.save_for_backward(x1, rowmax, rowmean, colmax, colmean, mean, s, g1, g2, saved, *params)
        fctx.meta = (B, H, W, bn_train, groups)
        return out

    @staticmethod
    def backward(fctx, dout):
        x1, rowmax, rowmean, colmax, colmean, mean, s, g1, g2, saved, *params = fctx.saved_tensors
        B, H, W, bn_train, groups = fctx.meta
        ctx = _ctx(dout.device)
        lib = _lib.lib()
        c = x1.shape[1]
        dev = dout.device
        dout = dout.contiguous()
        # gradients of the gates: sums of dOut * x1 over x, over y, over the map
        dg1, dg2, ds = torch.empty(B, H, c, device=dev), torch.empty(B, W, c, device=dev), torch.empty(B, c, device=dev)
        ws = torch.empty(B * lib.spei_plane_ws_floats(H, W, c), device=dev)
        _lib.check(lib.spei_plane_stats_batched(_p(ctx, dout), _p(ctx, x1), 1, H, W, c, _NULL, _p(ctx, dg1), _NULL, _p(ctx, dg2), _p(ctx, ds),
                                                _p(ctx, ws), B, ctx._stream()), "spei_plane_stats_batched")
        # ... through the gate maps: statistics and parameters are the leaves
        prm, run, keep_alive = _gate_ptrs(ctx, params)
        d_stats = [torch.empty_like(t) for t in (rowmax, rowmean, colmax, colmean, mean)]
        npar = lib.spei_gate_train_nparams(c)
        dprm = torch.empty(npar, device=dev)
        gws = torch.empty(lib.spei_gate_train_ws_floats(B, groups, H, W, c) // 2 + 1, device=dev, dtype=torch.float64)
        _lib.check(lib.spei_gate_maps_bwd(_p(ctx, rowmax), _p(ctx, rowmean), _p(ctx, colmax), _p(ctx, colmean), _p(ctx, mean), prm, run, B, groups,
                                          H, W, c, int(bn_train), _p(ctx, s), _p(ctx, saved), _p(ctx, ds), _p(ctx, dg1), _p(ctx, dg2),
                                          *[_p(ctx, t) for t in d_stats], _p(ctx, dprm), C.c_void_p(gws.data_ptr()), ctx._stream()),
                   "spei_gate_maps_bwd")
        dx1 = torch.empty_like(x1)
        _lib.check(lib.spei_resblock_apply_bwd_batched(_p(ctx, dout), _p(ctx, x1), _p(ctx, s), _p(ctx, g1), _p(ctx, g2), _p(ctx, rowmax),
                                                       _p(ctx, colmax), _p(ctx, d_stats[0]), _p(ctx, d_stats[1]), _p(ctx, d_stats[2]),
                                                       _p(ctx, d_stats[3]), _p(ctx, d_stats[4]), _p(ctx, dx1), B, H, W, c, ctx._stream()),
                   "spei_resblock_apply_bwd_batched")
        # dprm: se_w1 | se_b1 | se_w2 | se_b2 | cw_w | cw_g | cw_b | hc_w | hc_g | hc_b  -> the parameters' shapes (views of one buffer)
        r = c // 4
        sizes = (r * c, r, c * r, c, 98, 1, 1, 50, 1, 1)
        parts = torch.split(dprm, sizes)
        gi = iter(parts)
        dparams = []
        for i, t in enumerate(params):
            if i in (7, 8, 12, 13):
                dparams.append(None)                       # running buffers
            else:
                g = next(gi)
                dparams.append(g.view(t.shape) if fctx.needs_input_grad[7 + i] else None)
        return (dout, dx1, None, None, None, None, None) + tuple(dparams)


# ---- the model, train-mode graph ----------------------------------------------------------------------------------------------
def resblock(x: torch.Tensor, blk, B: int, H: int, W: int, bn_train: bool, groups: int = 1) -> torch.Tensor:
    """x + SE(x1) + TE(x1), x1 = conv5(relu(conv5(x)))   (model/block.py:127-140); `blk`: a speinet._ResBlock."""
    c0, c1 = blk.main[0].main[0], blk.main[1].main[0]
    t = _Conv2d.apply(x, c0.weight, c0.bias, None, B, H, W, 5, 1, True)
    x1 = _Conv2d.apply(t, c1.weight, c1.bias, None, B, H, W, 5, 1, False)
    cw, hc = blk.te.cw.conv, blk.te.hc.conv
    if bn_train:
        for bn in (cw.bn, hc.bn):
            bn.num_batches_tracked += groups
    return _GatedSum.apply(x, x1, B, H, W, bn_train, groups, blk.se.fc[0].weight, blk.se.fc[0].bias, blk.se.fc[2].weight, blk.se.fc[2].bias,
                           cw.conv.weight, cw.bn.weight, cw.bn.bias, cw.bn.running_mean, cw.bn.running_var,
                           hc.conv.weight, hc.bn.weight, hc.bn.bias, hc.bn.running_mean, hc.bn.running_var)


def encoder(frames: torch.Tensor, rn, bn_train: bool, pyramid: bool = False, groups: int = 1):
    """encoder_second(encoder_first(inBlock(frames)))  (model/swint.py:53,58): [B,3,H,W] -> [B*(H/4)*(W/4), 128]; pyramid: all three
    levels (lv1 [B*H*W, 32], lv2, lv3), the sharp-reference features of model/speinet.py:124-126.  groups: `frames` holds that many
    encoder passes of the reference one after the other (B / groups samples each, in the reference's call order): one set of
    launches, every parameter used once per step, the BatchNorm(1) statistics of the gates taken and their running buffers moved pass
    by pass as the separate calls would."""
    B, _, H, W = frames.shape
    assert B % groups == 0
    f = _ConvIn.apply(frames, rn.inBlock[0][0].weight, rn.inBlock[0][0].bias)
    for blk in list(rn.inBlock)[1:]:
        f = resblock(f, blk, B, H, W, bn_train, groups)
    levels = [f]
    h, w = H, W
    for stage in (rn.encoder_first, rn.encoder_second):
        f = _Conv2d.apply(f, stage[0][0].weight, stage[0][0].bias, None, B, h, w, 5, 2, True)
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        for blk in list(stage)[1:]:
            f = resblock(f, blk, B, h, w, bn_train, groups)
        levels.append(f)
    return tuple(levels) if pyramid else f


def dec_stage(f: torch.Tensor, stage, B: int, h: int, w: int, bn_train: bool) -> torch.Tensor:
    """n ResBlocks then ConvTranspose2d + ReLU (decoder_second / decoder_first, model/recons_video_ori.py:58-71): -> 2h x 2w."""
    for blk in list(stage)[:-1]:
        f = resblock(f, blk, B, h, w, bn_train)
    ct = stage[-1][0]
    return _ConvT2d.apply(f, ct.weight, ct.bias, B, h, w)


def out_block(f: torch.Tensor, rn, B: int, h: int, w: int, bn_train: bool) -> torch.Tensor:
    """outBlock (model/recons_video_ori.py:73-77): [B*h*w, 32] -> [B, 3, h, w]."""
    for blk in list(rn.outBlock)[:-1]:
        f = resblock(f, blk, B, h, w, bn_train)
    last = rn.outBlock[-1]
    # 32 -> 3 channels: the GEMM family works in 32-channel tiles, so the weights are zero-padded to 32 outputs (autograd
    # slices the gradient back); the three planes are a view of the first three columns
    w32 = F.pad(last.weight, (0, 0, 0, 0, 0, 0, 0, 32 - last.weight.shape[0]))
    b32 = F.pad(last.bias, (0, 32 - last.bias.shape[0]))
    o = _Conv2d.apply(f, w32, b32, None, B, h, w, 5, 1, False)
    return o.view(B, h, w, 32)[..., :last.weight.shape[0]].permute(0, 3, 1, 2).contiguous()


def decoder(f: torch.Tensor, rn, B: int, h: int, w: int, bn_train: bool) -> torch.Tensor:
    """outBlock(decoder_first(decoder_second(f)))  (model/swint.py:65): [B*h*w, 128] -> [B, 3, 4h, 4w]."""
    f = dec_stage(f, rn.decoder_second, B, h, w, bn_train)
    f = dec_stage(f, rn.decoder_first, B, 2 * h, 2 * w, bn_train)
    return out_block(f, rn, B, 4 * h, 4 * w, bn_train)


def drop_path_rates(depths: Sequence[int], drop_path_rate: float = 0.1) -> List[float]:
    """model/swinir.py:691: stochastic depth decay rule over all blocks."""
    return [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]


def drop_path_scales(depths: Sequence[int], B: int, n_calls: int, generator: Optional[torch.Generator] = None) -> list:
    """The DropPath factors of `n_calls` swin calls in the reference's call order: per call, per block with rate > 0, first the
    attention branch then the MLP branch, each `new_empty((B,1,1)).bernoulli_(keep) / keep` (timm.models.layers.DropPath,
    scale_by_keep) drawn on the CPU generator.  Returns [call][block] -> None or (attn [B], mlp [B])."""
    rates = drop_path_rates(depths)
    out = []
    for _ in range(n_calls):
        call = []
        for r in rates:
            if r <= 0.0:
                call.append(None)
                continue
            keep = 1.0 - r
            pair = tuple(torch.empty(B, 1, 1).bernoulli_(keep, generator=generator).div_(keep).view(B) for _ in range(2))
            call.append(pair)
        out.append(call)
    return out


def _rows(scale: Optional[torch.Tensor], hw: int, device) -> Optional[torch.Tensor]:
    return None if scale is None else scale.to(device=device, dtype=torch.float32).repeat_interleave(hw).contiguous()


def swin(sw, x: torch.Tensor, y: torch.Tensor, B: int, h: int, w: int, scales: Optional[list]) -> torch.Tensor:
    """SwinIR.forward(x, y) of the reference's cross-window-attention variant (model/swinir.py:781-810 with upsampler '',
    img_range 1, mean 0; :763-779 forward_features; :483-484 RSTB; :238-281 block): queries from y, keys / values from x.
    x, y [B*h*w, 128] -> [B*h*w, 128].  scales: one call's entry of `drop_path_scales` (None: no DropPath, the eval graph)."""
    hw = h * w
    cf = sw.conv_first
    x_first = _Conv2d.apply(x, cf.weight, cf.bias, None, B, h, w, 3, 1, False)
    y_first = _Conv2d.apply(y, cf.weight, cf.bias, None, B, h, w, 3, 1, False)
    pe = sw.patch_embed.norm
    xt = _LayerNorm.apply(x_first, pe.weight, pe.bias)
    yt = _LayerNorm.apply(y_first, pe.weight, pe.bias)
    bi = 0
    for layer in sw.layers:
        cur = xt
        for blk in layer.residual_group.blocks:
            sc = scales[bi] if scales is not None else None
            bi += 1
            shift = 0 if blk.attn_mask is None else 2
            at = blk.attn
            xn = _LayerNorm.apply(cur, blk.norm1.weight, blk.norm1.bias)
            yn = _LayerNorm.apply(yt, blk.norm1.weight, blk.norm1.bias)
            scale = (256 // 8) ** -0.5
            q = _Linear.apply(yn, at.qkv_y.weight * scale, at.qkv_y.bias * scale, None, None)
            kv = _Linear.apply(xn, at.qkv_x.weight, at.qkv_x.bias, None, None)
            relbias = at.relative_position_bias_table[at.relative_position_index.view(-1)].view(25, 25, -1).permute(2, 0, 1)
            a = _WindowAttention.apply(q, kv, relbias, B, h, w, shift)
            cur = _Linear.apply(a, at.proj.weight, at.proj.bias, cur, _rows(sc[0] if sc else None, hw, x.device))
            hn = _LayerNorm.apply(cur, blk.norm2.weight, blk.norm2.bias)
            hid = _Gelu.apply(_Linear.apply(hn, blk.mlp.fc1.weight, blk.mlp.fc1.bias, None, None))
            cur = _Linear.apply(hid, blk.mlp.fc2.weight, blk.mlp.fc2.bias, cur, _rows(sc[1] if sc else None, hw, x.device))
        xt = _Conv2d.apply(cur, layer.conv.weight, layer.conv.bias, xt, B, h, w, 3, 1, False)        # RSTB: conv(blocks(x)) + x
    xt = _LayerNorm.apply(xt, sw.norm.weight, sw.norm.bias)
    res = _Conv2d.apply(xt, sw.conv_after_body.weight, sw.conv_after_body.bias, x_first, B, h, w, 3, 1, False)
    return _Conv2d.apply(res, sw.conv_last.weight, sw.conv_last.bias, x, B, h, w, 3, 1, False)


def forward_swint(model, x: torch.Tensor, scales: Optional[list] = None, bn_train: Optional[bool] = None) -> torch.Tensor:
    """model/swint.py:51-67 as a differentiable graph on the HIP kernels.  x [B, >= n, 3, H, W]; `scales`: the DropPath factors
    (`drop_path_scales`; drawn here from torch's CPU generator when the model is in train() mode and none are given)."""
    if not x.is_cuda:
        raise RuntimeError("speinet_amd.train runs on MI355X only (HIP kernels); there is no CPU path")
    _lib.lib()
    n = model.n_sequence
    B, _, _, H, W = x.shape
    if H % 20 or W % 20:
        raise ValueError(f"H and W must be multiples of 20 (two stride-2 stages, then 5x5 windows); got {H}x{W}")
    training = model.training if bn_train is None else bn_train
    n_calls = 1 if n == 1 else n - 1
    if scales is None and model.training:
        scales = drop_path_scales(model.cfg.depths, B, n_calls)
    h3, w3 = H // 4, W // 4
    x = x.float()
    token = _PREC.set(_train_precision(model))
    try:
        return _forward_swint_graph(model, x, n, B, h3, w3, training, scales)
    finally:
        _PREC.reset(token)


def _train_precision(model) -> str:
    p = getattr(model, "train_precision", "f32")
    if p not in ("f32", "bf16x3"):
        raise ValueError(f"train_precision {p!r}: 'f32' or 'bf16x3'")
    return p


def merge_scales(calls: Optional[list]) -> Optional[list]:
    """The DropPath factors of several Swin calls (`drop_path_scales`: [call][block] -> None or (attn [B], mlp [B])) as those of ONE call
    on the calls' samples stacked: [block] -> None or (attn [calls*B], mlp [calls*B])."""
    if not calls:
        return None
    out = []
    for per_block in zip(*calls):
        if per_block[0] is None:
            out.append(None)
        else:
            out.append(tuple(torch.cat([c[j] for c in per_block]) for j in range(2)))
    return out


def _forward_swint_graph(model, x, n, B, h3, w3, training, scales):
    """The n encoder passes as one (groups = n, the centre frame first: the reference's call order, model/swint.py:53,58) and the
    n - 1 Swin calls as one call on the neighbours' samples stacked (the same weights; keys / values from the centre frame's
    features repeated): every parameter enters the graph once, so autograd has no per-parameter sums left to take."""
    with torch.cuda.device(x.device):
        rn = model.recons_net
        hw = B * h3 * w3
        order = [n // 2] + [i for i in range(n) if i != n // 2]
        enc = encoder(torch.cat([x[:, i] for i in order]), rn, training, groups=n)
        f_mid = enc[:hw]
        if n == 1:
            fused = f_mid + swin(model.swin, f_mid, f_mid, B, h3, w3, scales[0] if scales else None)
        else:
            calls = n - 1
            sw = swin(model.swin, f_mid.repeat(calls, 1), enc[hw:], calls * B, h3, w3, merge_scales(scales))
            fused = torch.cat([f_mid] + [sw[j * hw:(j + 1) * hw] for j in range(calls)], dim=1)
        cv = model.conv
        ff = _Linear.apply(fused, cv.weight.view(cv.weight.shape[0], -1), cv.bias, None, None)
        return decoder(ff, rn, B, h3, w3, training)


# ---- model/speinet.py, train-mode graph (trainer/trainer_swint_hsa_nsf.py) --------------------------------------------------------
def rl_prior(frames: torch.Tensor, iters: int, lam: float = 0.01) -> torch.Tensor:
    """r_l_per_channel(frame, 5x5 box, iters, lam) (model/rcl.py:22-51) on a batch of frames; data only, no gradient."""
    ctx = _ctx(frames.device)
    lib = _lib.lib()
    frames = frames.detach().contiguous()
    B, c, H, W = frames.shape
    out, scratch = torch.empty_like(frames), torch.empty(c, H, W, device=frames.device)
    for i in range(B):
        _lib.check(lib.spei_rl_prior(_p(ctx, frames[i]), _p(ctx, out[i]), _p(ctx, scratch), c, H, W, iters, lam, ctx._stream()), "spei_rl_prior")
    return out


def _conv1x1_relu(x, conv, B, h, w):
    return _Conv2d.apply(x, conv.weight, conv.bias, None, B, h, w, 1, 1, True)


def decode_speinet(m, ff, S, t3, t2, t1, B: int, h3: int, w3: int, bn_train: bool) -> torch.Tensor:
    """SPEINet._decode (model/speinet.py:92-120) on pixel rows: ff [B*h3*w3, 128], S [B*h3*w3], t3 / t2 / t1 the transferred
    features at the three levels."""
    rn = m.recons_net
    lin = lambda x, conv: _Linear.apply(x, conv.weight.view(conv.weight.shape[0], -1), conv.bias, None, None)
    f_lv3 = ff + _RowScale.apply(lin(torch.cat((ff, t3), dim=1), m.conv_lv3), S)
    dec2 = dec_stage(f_lv3, rn.decoder_second, B, h3, w3, bn_train)
    h2, w2 = 2 * h3, 2 * w3
    s_col = S.view(-1, 1)
    s2 = _Bicubic.apply(s_col, B, h3, w3, 2).view(-1)
    f_lv2 = dec2 + _RowScale.apply(lin(torch.cat((dec2, t2), dim=1), m.conv_lv2), s2)
    search_1 = _conv1x1_relu(_Bicubic.apply(f_lv3, B, h3, w3, 2), m.search1, B, h2, w2)
    search_2 = _Conv2d.apply(f_lv2, m.search3.weight, m.search3.bias, None, B, h2, w2, 3, 1, True)
    search_11 = _conv1x1_relu(torch.cat((dec2, search_1), dim=1), m.search2, B, h2, w2)
    search_22 = _conv1x1_relu(torch.cat((f_lv2, search_2), dim=1), m.search2, B, h2, w2)
    f_v3 = dec2 + search_11
    f_lv2 = f_lv2 + search_22
    dec1 = dec_stage(f_lv2, rn.decoder_first, B, h2, w2, bn_train)
    h1, w1 = 2 * h2, 2 * w2
    s4 = _Bicubic.apply(s_col, B, h3, w3, 4).view(-1)
    f_lv1 = dec1 + _RowScale.apply(lin(torch.cat((dec1, t1), dim=1), m.conv_lv1), s4)
    search_13 = _conv1x1_relu(_Bicubic.apply(f_v3, B, h2, w2, 2), m.search13, B, h1, w1)
    c33 = lambda x: _Conv2d.apply(x, m.search33.weight, m.search33.bias, None, B, h1, w1, 3, 1, True)
    search_23 = c33(_Bicubic.apply(f_lv2, B, h2, w2, 2))
    search_33 = _Conv2d.apply(f_lv1, m.search43.weight, m.search43.bias, None, B, h1, w1, 3, 1, True)
    f_lv1 = f_lv1 + c33(torch.cat((search_13, search_23), dim=1)) + c33(torch.cat((search_13, search_33), dim=1)) \
        + c33(torch.cat((search_23, search_33), dim=1))
    return out_block(f_lv1, rn, B, h1, w1, bn_train)


def _branch_speinet(m, x: torch.Tensor, has_ref: bool, scales: Optional[list], bn_train: bool) -> torch.Tensor:
    """SPEINet._forwardbs (has_ref) / _forwardb (model/speinet.py:122-148) for the samples of one routing class."""
    n = m.n_sequence
    B, _, _, H, W = x.shape
    h3, w3 = H // 4, W // 4
    rn = m.recons_net
    # every encoder pass of the branch as ONE pass, in the reference's call order (model/speinet.py:124-131 / :141-147): the sharp
    # reference (all three levels), the centre frame and its 5-step prior, then each neighbour and its 1-step prior
    mid = x[:, n // 2]
    nb = [i for i in range(n) if i != n // 2]
    passes = ([x[:, n + 1]] if has_ref else []) + [mid, rl_prior(mid, 5)]
    for i in nb:
        passes += [x[:, i], rl_prior(x[:, i], 1)]
    g = len(passes)
    lv1, lv2, lv3 = encoder(torch.cat(passes), rn, bn_train, pyramid=True, groups=g)
    hw = B * h3 * w3
    part = lambda t, j: t[j * (t.shape[0] // g):(j + 1) * (t.shape[0] // g)]
    lv = (part(lv1, 0), part(lv2, 0), part(lv3, 0)) if has_ref else None                 # sharp_lv1, sharp_lv2, sharp_lv3
    o = 1 if has_ref else 0
    f_mid = part(lv3, o) + part(lv3, o + 1)
    if nb:
        calls = len(nb)
        feat = torch.cat([part(lv3, o + 2 + 2 * j) + part(lv3, o + 3 + 2 * j) for j in range(calls)])
        sw = swin(m.swin, f_mid.repeat(calls, 1), feat, calls * B, h3, w3, merge_scales(scales))
        feats = [f_mid] + [sw[j * hw:(j + 1) * hw] for j in range(calls)]
    else:
        feats = [f_mid]
    fu = m.fusion
    ff = _Linear.apply(torch.cat(feats, dim=1), fu.weight.view(fu.weight.shape[0], -1), fu.bias, None, None)
    if has_ref:
        S, t3, t2, t1, _ = _SearchTransfer.apply(ff, lv[2], lv[1], lv[0], B, h3, w3, h3, w3)
    else:
        # SelfTransfer: the reference map is the query map rotated (x.transpose(2,3).flip(2), model/SearchTransfer.py:60): a pure
        # re-indexing, done with tensor views; only S comes out of the search, the transfers are up-sampled convs of ff
        ref = ff.view(B, h3, w3, -1).transpose(1, 2).flip(1).reshape(B * h3 * w3, -1)
        S, _, _, _, _ = _SearchTransfer.apply(ff, ref, None, None, B, h3, w3, w3, h3)
        st = m.SelfTransfer
        t3 = ff
        t2 = _conv1x1_relu(_Bicubic.apply(ff, B, h3, w3, 2), st.search1, B, 2 * h3, 2 * w3)
        t1 = _conv1x1_relu(_Bicubic.apply(t2, B, 2 * h3, 2 * w3, 2), st.search2, B, 4 * h3, 4 * w3)
    return decode_speinet(m, ff, S, t3, t2, t1, B, h3, w3, bn_train)


def speinet_drop_path_scales(depths, zero_ref: Sequence[bool], n_sequence: int, generator: Optional[torch.Generator] = None) -> dict:
    """DropPath factors in the order model/speinet.py:150-168 consumes them: first the no-reference samples' branch (`_forwardb`),
    then the others' (`_forwardbs`); each runs n_sequence - 1 swin calls on its own sub-batch.  {has_ref: scales}."""
    out = {}
    for has_ref in (False, True):
        nb = sum(1 for z in zero_ref if z != has_ref)
        if nb:
            out[has_ref] = drop_path_scales(depths, nb, n_sequence - 1, generator)
    return out


def forward_speinet(model, x: torch.Tensor, scales: Optional[dict] = None, bn_train: Optional[bool] = None) -> torch.Tensor:
    """model/speinet.py:150-168 as a differentiable graph on the HIP kernels: x [B, n+2, 3, H, W]; samples whose frame 3 is all
    zero go through `_forwardb` (SelfTransfer), the others through `_forwardbs` (SearchTransfer), each class as one sub-batch
    (BatchNorm statistics per class, as in the reference)."""
    if not x.is_cuda:
        raise RuntimeError("speinet_amd.train runs on MI355X only (HIP kernels); there is no CPU path")
    _lib.lib()
    B, _, _, H, W = x.shape
    if H % 20 or W % 20:
        raise ValueError(f"H and W must be multiples of 20 (two stride-2 stages, then 5x5 windows); got {H}x{W}")
    training = model.training if bn_train is None else bn_train
    x = x.float()
    zero_ref = [bool(v) for v in (x[:, 3].reshape(B, -1) == 0).all(dim=1).tolist()]
    if scales is None and model.training:
        scales = speinet_drop_path_scales(model.cfg.depths, zero_ref, model.n_sequence)
    out = torch.zeros(B, 3, H, W, device=x.device)
    token = _PREC.set(_train_precision(model))
    try:
        with torch.cuda.device(x.device):
            for has_ref in (False, True):
                idx = torch.tensor([i for i, z in enumerate(zero_ref) if z != has_ref], device=x.device, dtype=torch.long)
                if idx.numel():
                    o = _branch_speinet(model, x[idx].contiguous(), has_ref, scales.get(has_ref) if scales else None, training)
                    out = out.index_copy(0, idx, o)
    finally:
        _PREC.reset(token)
    return out
