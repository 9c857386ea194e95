"""Thin tensor-level wrappers over the C-ABI (include/speinet_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic step is one of the HIP
kernels in speinet_amd/csrc.  There is no fallback: a missing library or a failed call raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .pack import PackedW

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
CONV, CONV_T = 0, 1

# Arithmetic of the GEMM-shaped kernels (convolutions, linears, correlation); everything else is always fp32.
#   "f32"    v_mfma_f32_32x32x2_f32, exact fp32 (the PSNR-parity configuration)
#   "bf16x3" split-bf16 products on v_mfma_f32_32x32x16_bf16: f32-grade results at 3/16 of the f32 MFMA cost
#   "bf16"   single bf16 products, fp32 accumulate (the throughput configuration of BASELINE.json configs[1])
# CORR_PRECISION applies to the correlation arg-max when PRECISION != "f32" ("bf16x3" keeps the arg-max stable).
PRECISION = "f32"
CORR_PRECISION = "bf16x3"
USE_SLAB = True      # bf16 modes: slab-resident conv/linear kernel (conv_slab_bf16.hip) instead of igemm_bf16.hip
BF16_STORAGE = True  # "bf16" mode: tensors that only feed the next GEMM / the attention kernel live in HBM as bf16


def inter_dtype() -> torch.dtype:
    """Storage type of GEMM-only intermediates (x-hat, q, kv, attention output, MLP hidden, ResBlock conv1 output)."""
    return torch.bfloat16 if (PRECISION == "bf16" and USE_SLAB and BF16_STORAGE) else torch.float32


def set_precision(mode: str, corr: str = None) -> None:
    global PRECISION, CORR_PRECISION
    if mode not in ("f32", "bf16x3", "bf16"):
        raise ValueError(f"unknown precision {mode!r}")
    PRECISION = mode
    if corr is not None:
        if corr not in ("bf16x3", "bf16"):
            raise ValueError(f"unknown correlation precision {corr!r}")
        CORR_PRECISION = corr

# Optional per-op timing hook for bench.py: {op name: [(start_event, end_event), ...]} recorded on the
# current stream (the stream the kernels are launched on).  None = off (default).
PROFILE = None


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PROFILE is not None and self.name in PROFILE:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if PROFILE is not None and self.name in PROFILE:
            self.e.record()
            PROFILE[self.name].append((self.s, self.e))
        return False


class FMap:
    """NHWC feature map view (fp32, or bf16 for GEMM-only intermediates): rows = pixels, `C` channels starting at
    column `off` of a [H*W, ld] buffer."""
    __slots__ = ("t", "H", "W", "C", "ld", "off")

    def __init__(self, t: torch.Tensor, H: int, W: int, C_: int, off: int = 0):
        assert t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous() and t.dim() == 2 and t.shape[0] == H * W
        self.t, self.H, self.W, self.C, self.ld, self.off = t, H, W, C_, t.shape[1], off
        assert off + C_ <= self.ld

    @staticmethod
    def empty(H: int, W: int, C_: int, device, dtype=torch.float32) -> "FMap":
        return FMap(torch.empty(H * W, C_, device=device, dtype=dtype), H, W, C_)

    @property
    def bf16(self) -> bool:
        return self.t.dtype == torch.bfloat16

    def view(self, off: int, C_: int) -> "FMap":
        return FMap(self.t, self.H, self.W, C_, self.off + off)

    @property
    def ptr(self) -> int:
        return self.t.data_ptr() + self.t.element_size() * self.off

    def dense(self) -> torch.Tensor:
        """[H, W, C] copy-free when the view spans the whole buffer."""
        return self.t[:, self.off:self.off + self.C].float().reshape(self.H, self.W, self.C)

    def nchw(self) -> torch.Tensor:
        return self.dense().permute(2, 0, 1).unsqueeze(0).contiguous()

    @staticmethod
    def from_nchw(x: torch.Tensor) -> "FMap":
        """[1,C,H,W] or [C,H,W] -> NHWC map (test helper; the forward pass itself never converts through torch)."""
        if x.dim() == 4:
            x = x[0]
        c, h, w = x.shape
        return FMap(x.permute(1, 2, 0).reshape(h * w, c).contiguous().float(), h, w, c)


def _vp(p) -> C.c_void_p:
    return C.c_void_p(p)


def _tp(t: Optional[torch.Tensor]) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def any_nonzero(x: torch.Tensor, flag: torch.Tensor) -> None:
    _lib.check(_lib.lib().spei_any_nonzero(_tp(x), x.numel(), _tp(flag), _stream()), "spei_any_nonzero")


def rl_prior(img: torch.Tensor, iters: int, lam: float = 0.01) -> torch.Tensor:
    """img [3,H,W] -> [3,H,W]."""
    c, h, w = img.shape
    out = torch.empty_like(img)
    scratch = torch.empty_like(img)
    _lib.check(_lib.lib().spei_rl_prior(_tp(img), _tp(out), _tp(scratch), c, h, w, iters, lam, _stream()), "spei_rl_prior")
    return out


def conv5_in(img: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> FMap:
    c, h, wd = img.shape
    assert c == 3
    out = FMap.empty(h, wd, b.numel(), img.device)
    _lib.check(_lib.lib().spei_conv5_in(_tp(img), _tp(w), _tp(b), _vp(out.ptr), h, wd, b.numel(), _stream()), "spei_conv5_in")
    return out


def conv5_out(f: FMap, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, w32=None, b32: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Last conv, NHWC 32 channels -> three NCHW planes.  w32 / b32 (weights zero-padded to 32 output channels, packed):
    the "bf16" mode runs the layer on the slab kernel."""
    assert out.shape == (3, f.H, f.W) and out.is_contiguous() and out.dtype == torch.float32
    if PRECISION == "bf16" and USE_SLAB and w32 is not None and f.C == 32:
        _lib.check(_lib.lib().spei_conv5_out_slab_bf16(_vp(f.ptr), f.ld, int(f.bf16), _tp(w32.fhi), _tp(b32), _tp(out), f.H, f.W, _stream()),
                   "spei_conv5_out_slab_bf16")
        return out
    assert not f.bf16
    _lib.check(_lib.lib().spei_conv5_out(_vp(f.ptr), f.ld, _tp(w), _tp(b), _tp(out), f.H, f.W, f.C, _stream()), "spei_conv5_out")
    return out


def igemm(a0: FMap, w: torch.Tensor, bias: Optional[torch.Tensor], N: int, ksize: int = 1, stride: int = 1,
          mode: int = CONV, act: int = ACT_NONE, a1: Optional[FMap] = None, residual: Optional[FMap] = None,
          rowscale: Optional[torch.Tensor] = None, out: Optional[FMap] = None, out_dtype=torch.float32,
          ln_input: bool = False) -> FMap:
    pad = ksize // 2
    if mode == CONV:
        ho, wo = (a0.H + 2 * pad - ksize) // stride + 1, (a0.W + 2 * pad - ksize) // stride + 1
    else:
        ho, wo = a0.H * stride, a0.W * stride
    if out is None:
        out = FMap.empty(ho, wo, N, a0.t.device, out_dtype)
    assert out.H == ho and out.W == wo and out.C == N
    k0, k1 = a0.C, (a1.C if a1 is not None else 0)
    slab = PRECISION != "f32" and USE_SLAB and mode == CONV
    assert slab or (mode == CONV_T and PRECISION == "bf16" and USE_SLAB) or not (a0.bf16 or out.bf16), \
        "bf16 activations are only supported by the slab kernel"
    assert a1 is None or a1.bf16 == a0.bf16
    assert residual is None or not residual.bf16
    assert not ln_input or (slab and w.fhi is not None), "ln_input is a feature of the slab kernel"
    if torch.is_tensor(w):
        w = PackedW(w, a0.t.device)
    assert tuple(w.shape) == (ksize * ksize, N, k0 + k1), (tuple(w.shape), ksize, N, k0, k1)
    if a1 is not None:
        assert (a1.H, a1.W) == (a0.H, a0.W)
    if residual is not None:
        assert (residual.H, residual.W, residual.C) == (ho, wo, N)
    if rowscale is not None:
        assert rowscale.numel() == ho * wo
    common = (_vp(out.ptr), out.ld, _vp(residual.ptr if residual is not None else 0),
              residual.ld if residual is not None else 0, _tp(rowscale), a0.H, a0.W, ho, wo, N, ksize, stride, pad, mode, act,
              _stream())
    srcs = (_vp(a0.ptr), a0.ld, k0, _vp(a1.ptr if a1 is not None else 0), a1.ld if a1 is not None else 0, k1)
    if (mode == CONV_T and PRECISION == "bf16" and USE_SLAB and ksize == 3 and stride == 2 and a1 is None and residual is None
            and rowscale is None and N % 32 == 0 and k0 % 32 == 0):
        # stride-2 transposed conv = four stride-1 convs (one per output parity) on the slab kernel
        cf = w.convT_class_frags()
        _lib.check(_lib.lib().spei_convt2_slab_bf16(_vp(a0.ptr), a0.ld, k0, int(a0.bf16), _tp(cf[(0, 0)]), _tp(cf[(0, 1)]), _tp(cf[(1, 0)]),
                                                   _tp(cf[(1, 1)]), _tp(bias), _vp(out.ptr), out.ld, int(out.bf16), a0.H, a0.W, N, act,
                                                   _stream()), "spei_convt2_slab_bf16")
    elif PRECISION == "f32":
        _lib.check(_lib.lib().spei_igemm_f32(*srcs, _tp(w.f32), _tp(bias), *common), "spei_igemm_f32")
    elif slab and w.fhi is not None:
        dims = (a0.H * a0.W, 1, ho * wo, 1) if (ksize == 1 and stride == 1) else (a0.H, a0.W, ho, wo)
        _lib.check(_lib.lib().spei_conv_slab_bf16(
            *srcs, int(a0.bf16), _tp(w.fhi), _tp(w.flo) if PRECISION == "bf16x3" else _vp(0), _tp(bias), _vp(out.ptr), out.ld,
            int(out.bf16), _vp(residual.ptr if residual is not None else 0), residual.ld if residual is not None else 0,
            _tp(rowscale), *dims, N, ksize, stride, pad, act, int(ln_input), _stream()), "spei_conv_slab_bf16")
    else:
        _lib.check(_lib.lib().spei_igemm_bf16(*srcs, _tp(w.hi), _tp(w.lo) if PRECISION == "bf16x3" else _vp(0), _tp(bias), *common),
                   "spei_igemm_bf16")
    return out


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int = ACT_NONE,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_dtype=torch.float32,
           ln_input: bool = False) -> torch.Tensor:
    """Token-space linear: x [M,K] -> [M,N]; w [N,K]."""
    m, k = x.shape
    if torch.is_tensor(w):
        w = PackedW(w.reshape(1, *w.shape), x.device)
    n = w.shape[1]
    if out is None:
        out = torch.empty(m, n, device=x.device, dtype=out_dtype)
    igemm(FMap(x, m, 1, k), w, b, n, act=act, ln_input=ln_input,
          residual=FMap(residual, m, 1, n) if residual is not None else None, out=FMap(out, m, 1, n))
    return out


def resblock_gates(x1: FMap, pk: dict):
    dev = x1.t.device
    lib = _lib.lib()
    assert x1.off == 0 and x1.ld == x1.C
    s = torch.empty(x1.C, device=dev)
    g1 = torch.empty(x1.H, x1.C, device=dev)
    g2 = torch.empty(x1.W, x1.C, device=dev)
    ws = torch.empty(lib.spei_gate_ws_floats(x1.H, x1.W, x1.C), device=dev)
    _lib.check(lib.spei_resblock_gates(_vp(x1.ptr), int(x1.bf16), x1.H, x1.W, x1.C, _tp(pk["se_w1"]), _tp(pk["se_b1"]), _tp(pk["se_w2"]),
                                       _tp(pk["se_b2"]), _tp(pk["cw_w"]), _tp(pk["cw_bn"]), _tp(pk["hc_w"]), _tp(pk["hc_bn"]),
                                       _tp(s), _tp(g1), _tp(g2), _tp(ws), _stream()), "spei_resblock_gates")
    return s, g1, g2


def resblock(x: FMap, pk: dict, extra: Optional[FMap] = None, out: Optional[FMap] = None) -> FMap:
    """x + SE(x1) + TE(x1), x1 = conv5(relu(conv5(x)))  (reference model/block.py:127-140)."""
    c = x.C
    assert x.off == 0 and x.ld == c
    t = igemm(x, pk["w1"], pk["b1"], c, ksize=5, act=ACT_RELU, out_dtype=inter_dtype())   # only conv2 reads it
    x1 = igemm(t, pk["w2"], pk["b2"], c, ksize=5, out_dtype=inter_dtype() if X1_BF16 else torch.float32)
    s, g1, g2 = resblock_gates(x1, pk)
    if out is None:
        out = FMap.empty(x.H, x.W, c, x.t.device)
    if extra is not None:
        assert extra.off == 0 and extra.ld == c
    _lib.check(_lib.lib().spei_resblock_apply(_vp(x.ptr), _vp(x1.ptr), int(x1.bf16), _tp(s), _tp(g1), _tp(g2),
                                              _vp(extra.ptr if extra is not None else 0), _vp(out.ptr), out.ld, x.H, x.W, c,
                                              _stream()), "spei_resblock_apply")
    return out


X1_BF16 = True       # "bf16" mode: the ResBlock's conv2 output (read by the gate statistics and the apply pass) is bf16
FUSE_MLP = True      # "bf16" mode: LayerNorm -> fc1 -> GELU -> fc2 -> +x in one kernel (mlp_fused_bf16.hip)


def ln_fused_available() -> bool:
    """LayerNorm folded into the staging of the following 256-wide linear (slab kernel, any bf16 mode)."""
    return PRECISION != "f32" and USE_SLAB


FUSE_ATTN = True     # "bf16" mode: LayerNorm -> q/kv GEMMs -> window attention -> proj -> +x in one kernel


def attn_fused_available() -> bool:
    return PRECISION == "bf16" and USE_SLAB and FUSE_ATTN and BF16_STORAGE


def attn_fused(x: torch.Tensor, yhat: torch.Tensor, bk: dict, H: int, W: int, shift: int, out: torch.Tensor) -> torch.Tensor:
    """out = x + proj(window_attention(...))  (reference model/swinir.py:238-278); in place when out is x."""
    assert x.shape == (H * W, 256) and x.dtype == torch.float32 and out.shape == x.shape and out.dtype == torch.float32
    assert yhat.shape == x.shape and yhat.dtype == torch.bfloat16
    _lib.check(_lib.lib().spei_attn_fused_bf16(_tp(x), _tp(out), _tp(yhat), _tp(bk["wq"].fhi), _tp(bk["bq"]), _tp(bk["wkv"].fhi),
                                               _tp(bk["bkv"]), _tp(bk["wproj"].fhi), _tp(bk["bproj"]), _tp(bk["relbias"]), H, W, shift,
                                               _stream()), "spei_attn_fused_bf16")
    return out


def mlp_fused_available() -> bool:
    return PRECISION == "bf16" and USE_SLAB and FUSE_MLP


def mlp_fused(x: torch.Tensor, w1, b1: torch.Tensor, w2, b2: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out = x + fc2(gelu(fc1(LN(x))))  (reference model/swinir.py:279); in place when out is x."""
    assert x.shape[1] == 256 and x.dtype == torch.float32 and out.shape == x.shape and out.dtype == torch.float32
    assert tuple(w1.shape) == (1, 512, 256) and tuple(w2.shape) == (1, 256, 512)
    _lib.check(_lib.lib().spei_mlp_fused_bf16(_tp(x), _tp(out), _tp(w1.fhi), _tp(b1), _tp(w2.fhi), _tp(b2), x.shape[0], _stream()),
               "spei_mlp_fused_bf16")
    return out


def layernorm(x: torch.Tensor, g: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, out_dtype=torch.float32) -> torch.Tensor:
    assert x.shape[1] == 256 and x.is_contiguous() and x.dtype == torch.float32
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    _lib.check(_lib.lib().spei_layernorm256(_tp(x), _tp(out), int(out.dtype == torch.bfloat16), _tp(g), _tp(b), x.shape[0], _stream()),
               "spei_layernorm256")
    return out


def window_attention(q: torch.Tensor, kv: torch.Tensor, relbias: torch.Tensor, H: int, W: int, shift: int,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert q.shape == (H * W, 256) and kv.shape == (H * W, 512) and relbias.shape == (8, 25, 25)
    if out is None:
        out = torch.empty_like(q)
    assert q.dtype == kv.dtype == out.dtype
    _lib.check(_lib.lib().spei_window_attention(_tp(q), _tp(kv), int(q.dtype == torch.bfloat16), _tp(relbias), _tp(out), H, W, shift,
                                                _stream()), "spei_window_attention")
    return out


def patch_invnorm(f: FMap) -> torch.Tensor:
    inv = torch.empty(f.H * f.W, device=f.t.device)
    _lib.check(_lib.lib().spei_patch_invnorm(_vp(f.ptr), f.ld, _tp(inv), f.H, f.W, f.C, _stream()), "spei_patch_invnorm")
    return inv


def corr_argmax(lr: FMap, ref: FMap, inv_lr: torch.Tensor, inv_ref: torch.Tensor):
    lib = _lib.lib()
    dev = lr.t.device
    n = lr.H * lr.W
    s = torch.empty(n, device=dev)
    arg = torch.empty(n, device=dev, dtype=torch.int32)
    ws = torch.empty(lib.spei_corr_ws_floats(n), device=dev)
    if PRECISION == "f32":
        with _timed("corr_argmax"):
            _lib.check(lib.spei_corr_argmax(_vp(lr.ptr), lr.ld, _vp(ref.ptr), ref.ld, _tp(inv_lr), _tp(inv_ref), lr.H, lr.W, ref.H, ref.W,
                                            lr.C, _tp(s), _tp(arg), _tp(ws), _stream()), "spei_corr_argmax")
        return s, arg
    split = CORR_PRECISION == "bf16x3"
    parts = []
    for f in (lr, ref):
        hi = torch.empty(f.H * f.W, f.C, device=dev, dtype=torch.bfloat16)
        lo = torch.empty_like(hi) if split else None
        _lib.check(lib.spei_split_bf16(_vp(f.ptr), f.ld, _tp(hi), _tp(lo), f.H * f.W, f.C, _stream()), "spei_split_bf16")
        parts += [hi, lo]
    fn, name = (lib.spei_corr_slab_bf16, "spei_corr_slab_bf16") if (USE_SLAB and lr.C == 128) else (lib.spei_corr_argmax_bf16, "spei_corr_argmax_bf16")
    with _timed("corr_argmax"):
        _lib.check(fn(_tp(parts[0]), _tp(parts[1]), _tp(parts[2]), _tp(parts[3]), _tp(inv_lr), _tp(inv_ref),
                                             lr.H, lr.W, ref.H, ref.W, lr.C, _tp(s), _tp(arg), _tp(ws), _stream()), name)
    return s, arg


def gather_fold(ref: FMap, arg: torch.Tensor, H3: int, W3: int, Hr3: int, Wr3: int, s: int) -> FMap:
    assert ref.H == Hr3 * s and ref.W == Wr3 * s
    out = FMap.empty(H3 * s, W3 * s, ref.C, ref.t.device)
    _lib.check(_lib.lib().spei_gather_fold(_vp(ref.ptr), ref.ld, _tp(arg), _vp(out.ptr), out.ld, H3, W3, Hr3, Wr3, ref.C, s, _stream()),
               "spei_gather_fold")
    return out


def rot90(f: FMap) -> FMap:
    out = FMap.empty(f.W, f.H, f.C, f.t.device)
    _lib.check(_lib.lib().spei_rot90(_vp(f.ptr), f.ld, _vp(out.ptr), f.H, f.W, f.C, _stream()), "spei_rot90")
    return out


def upsample(f: FMap, s: int, act: int = ACT_NONE) -> FMap:
    out = FMap.empty(f.H * s, f.W * s, f.C, f.t.device)
    _lib.check(_lib.lib().spei_upsample_bicubic(_vp(f.ptr), f.ld, _vp(out.ptr), out.ld, f.H, f.W, f.C, s, act, _stream()),
               "spei_upsample_bicubic")
    return out


def up_conv1x1_relu(f: FMap, w, b: torch.Tensor, n: int, s: int = 2) -> FMap:
    """relu(conv1x1(bicubic_up(f))) (reference model/speinet.py:96-97,108-109, model/SearchTransfer.py:73-76).  Both maps
    are linear and the bicubic weights sum to 1, so the "bf16" mode runs the conv first, at 1/s^2 of the pixels and with
    half the bytes through the upsampler; the f32-grade modes keep the reference's order of operations."""
    if PRECISION == "bf16":
        return upsample(igemm(f, w, b, n), s, act=ACT_RELU)
    return igemm(upsample(f, s), w, b, n, act=ACT_RELU)


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    _lib.check(_lib.lib().spei_add(_tp(a), _tp(b), _tp(out), a.numel(), _stream()), "spei_add")
    return out
