"""Thin tensor-level wrappers over the C-ABI (include/speinet_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic step is one of the HIP
kernels in speinet_amd/csrc.  There is no fallback: a missing library or a failed call raises.

All state of a call lives in a `Ctx` (arithmetic mode, storage knobs, device, optional per-op timing hook): there is
NO process-global mode — two models with different modes can run from two threads (the reference's nn.DataParallel
calls `forward` from one Python thread per device, model/__init__.py:19-20; SURVEY.md §8b "threading").
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .pack import BF16, F16, F32, LP_DTYPE, PackedW, fmt_of

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
CONV, CONV_T = 0, 1

# Arithmetic of the GEMM-shaped kernels (convolutions, linears, correlation); everything else is always fp32.
#   "f32"    v_mfma_f32_32x32x2_f32, exact fp32 (the PSNR-parity configuration)
#   "bf16x3" split-bf16 products on v_mfma_f32_32x32x16_bf16: f32-grade results at 3/16 of the f32 MFMA cost
#   "bf16"   single bf16 products, fp32 accumulate (BASELINE.json configs[1] as written: 8-bit significands)
#   "f16"    single IEEE-half products, fp32 accumulate: the same matrix-pipe rate with 11-bit significands — the
#            throughput configuration that also holds the 1e-3 dB PSNR bound; range +-65504
# Correlation arg-max when precision != "f32" (16-bit products use the format of `precision`, bf16 unless "f16"):
#   "bf16x3" f32-grade scores;  "single" one 16-bit product per MAC (the winner may flip between near-ties);
#   "top2"   single products keeping the TWO best candidates of every query, then an exact re-score of both on the fp32
#            maps with fp64 accumulation (spei_corr_rescore): f32-grade winners and S at the cost of the 16-bit kernel
PRECISIONS = ("f32", "bf16x3", "bf16", "f16")
CORR_PRECISIONS = ("bf16x3", "single", "top2")
_CORR_ALIASES = {"bf16": "single", "bf16r": "top2", "f16": "single"}


class FMap:
    """NHWC feature map view (fp32, or bf16 / half for GEMM-only intermediates): rows = pixels, `C` channels starting at
    column `off` of a [H*W, ld] buffer."""
    __slots__ = ("t", "H", "W", "C", "ld", "off")

    def __init__(self, t: torch.Tensor, H: int, W: int, C_: int, off: int = 0):
        assert t.is_cuda and t.dtype in (torch.float32, torch.bfloat16, torch.float16) and t.is_contiguous() and t.dim() == 2 and t.shape[0] == H * W
        self.t, self.H, self.W, self.C, self.ld, self.off = t, H, W, C_, t.shape[1], off
        assert off + C_ <= self.ld

    @staticmethod
    def empty(H: int, W: int, C_: int, device, dtype=torch.float32) -> "FMap":
        return FMap(torch.empty(H * W, C_, device=device, dtype=dtype), H, W, C_)

    @property
    def lp(self) -> bool:
        """Stored as 16-bit (bf16 or half)."""
        return self.t.dtype != torch.float32

    @property
    def fmt(self) -> int:
        return fmt_of(self.t.dtype)

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


class BMap:
    """`B` dense NHWC maps of one shape stacked in one buffer [B*H*W, C] (fp32 or the mode's 16-bit format): the frame's encoder
    passes through one layer, launched together (gridDim.y = map; csrc/conv_slab16.hip, resblock.hip).  `map(b)` is map b as an FMap
    view."""
    __slots__ = ("t", "B", "H", "W", "C")

    def __init__(self, t: torch.Tensor, B: int, H: int, W: int, C_: int):
        assert t.is_cuda and t.is_contiguous() and t.shape == (B * H * W, C_)
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C_

    @staticmethod
    def empty(B: int, H: int, W: int, C_: int, device, dtype=torch.float32) -> "BMap":
        return BMap(torch.empty(B * H * W, C_, device=device, dtype=dtype), B, H, W, C_)

    @property
    def fmt(self) -> int:
        return fmt_of(self.t.dtype)

    def map(self, b: int) -> FMap:
        n = self.H * self.W
        return FMap(self.t[b * n:(b + 1) * n], self.H, self.W, self.C)


def _vp(p) -> C.c_void_p:
    return C.c_void_p(p)


class _timed:
    def __init__(self, ctx: "Ctx", name: str):
        self.on = ctx.profile is not None and name in ctx.profile
        self.ctx, self.name = ctx, name

    def __enter__(self):
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record(torch.cuda.current_stream(self.ctx.device))
        return self

    def __exit__(self, *a):
        if self.on:
            self.e.record(torch.cuda.current_stream(self.ctx.device))
            self.ctx.profile[self.name].append((self.s, self.e))
        return False


def _tensor_bytes(x) -> int:
    t = x.t if hasattr(x, "t") else x
    return t.numel() * t.element_size() if torch.is_tensor(t) else 0


def _conv_sub(method: str, a, k, out):
    """(sub-family, flops, activation bytes) of one conv-family launch: the 32-channel 5x5 layers at full resolution (the ResBlock convs of
    level 1 and the first / last conv) move 0.4-1.6 GB per launch for 0.3 TFLOP — their roofline is HBM, not the matrix pipe."""
    if method == "conv5_in":
        return "level1", 2.0 * 25 * 3 * 32 * a[0].shape[-2] * a[0].shape[-1], _tensor_bytes(a[0]) + _tensor_bytes(out)
    if method == "conv5_out":
        return "level1", 2.0 * 25 * 32 * 3 * a[0].H * a[0].W, _tensor_bytes(a[0]) + 12 * a[0].H * a[0].W
    src, n = a[0], a[3]
    ks = a[4] if len(a) > 4 else k.get("ksize", 1)
    if n == 32 and src.C == 32 and ks == 5 and k.get("stride", a[5] if len(a) > 5 else 1) == 1:
        maps = getattr(src, "B", 1)
        return "level1", 2.0 * 25 * 32 * 32 * maps * src.H * src.W, _tensor_bytes(src) + _tensor_bytes(out)
    return "rest", 0.0, 0


def _family(name: str):
    """Time every call of a leaf launch method when the caller asked for it: `profile["families"]` = {family: [(start, end), ...]} gets
    one HIP event pair per call on the launch stream (eager, single-stream passes only: bench.py's per-family roofline record); the
    conv family also by sub-family (`profile["conv_sub"]`: events, flops, bytes)."""
    def deco(fn):
        def wrapper(self, *a, **k):
            pr = self.profile
            if pr is None or "families" not in pr:
                return fn(self, *a, **k)
            st = torch.cuda.current_stream(self.device)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(st)
            out = fn(self, *a, **k)
            e_.record(st)
            pr["families"].setdefault(name, []).append((s_, e_))
            if name == "conv":
                sub, fl, by = _conv_sub(fn.__name__, a, k, out)
                rec = pr.setdefault("conv_sub", {}).setdefault(sub, {"events": [], "flops": 0.0, "bytes": 0})
                rec["events"].append((s_, e_))
                rec["flops"] += fl
                rec["bytes"] += by
            return out
        wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
        return wrapper
    return deco


def frame_post(out_chw: torch.Tensor, gt_hwc: torch.Tensor, border: int = 4):
    """Harness post-processing of one frame on the device (csrc/metrics.hip; reference inference_SPEINet.py:477-543): out_chw [3,H,W]
    fp32 -> (uint8 [H,W,3] frame, float64 [finite, PSNR, SSIM] against gt_hwc on the border-cropped region), three launches on the
    current stream of the tensors' device, no host sync."""
    assert out_chw.is_cuda and out_chw.dtype == torch.float32 and out_chw.dim() == 3 and out_chw.shape[0] == 3 and out_chw.is_contiguous()
    h, w = out_chw.shape[1:]
    assert gt_hwc.shape == (h, w, 3) and gt_hwc.dtype == torch.uint8 and gt_hwc.device == out_chw.device and gt_hwc.is_contiguous()
    lib = _lib.lib()
    n = lib.spei_frame_post_ws_doubles(h, w, border)
    if n < 0:
        raise ValueError(f"frame {w}x{h} with border {border} is smaller than the 11x11 SSIM window")
    dev = out_chw.device
    u8 = torch.empty(h, w, 3, dtype=torch.uint8, device=dev)
    ws = torch.empty(n, dtype=torch.float64, device=dev)
    res = torch.empty(3, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.spei_frame_post(_vp(out_chw.data_ptr()), _vp(gt_hwc.data_ptr()), _vp(u8.data_ptr()), h, w, border, _vp(ws.data_ptr()),
                                       _vp(res.data_ptr()), st), "spei_frame_post")
    return u8, res


CORR_DIAG_WS_MAX = 8 << 30      # bytes of candidate pairs the diagonal correlation kernel may use before the slab kernel takes over


class CorrPlan:
    """A prepared K11 launch (see Ctx.corr_plan): `launch()` runs the arg-max kernel — bracketed by the profile events — and
    whatever must follow it on the same stream (the exact re-score of the "top2" form)."""
    __slots__ = ("ctx", "s", "arg", "kernel", "main", "post", "keep")

    def __init__(self, ctx, s, arg, kernel, main, post, keep):
        self.ctx, self.s, self.arg, self.kernel, self.main, self.post, self.keep = ctx, s, arg, kernel, main, post, keep

    def launch(self, profile: Optional[dict] = None) -> None:
        ctx = self.ctx if profile is None else self.ctx.replace(profile=profile)
        st = ctx._stream()
        if ctx.profile is not None:
            ctx.profile["corr_kernel"] = self.kernel
        fam = ctx.profile is not None and "families" in ctx.profile
        if fam:
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(torch.cuda.current_stream(ctx.device))
        with _timed(ctx, "corr_argmax"):
            for fn, name, args in self.main:
                _lib.check(fn(*args, st), name)
        for fn, name, args in self.post:
            _lib.check(fn(*args, st), name)
        if fam:
            f1.record(torch.cuda.current_stream(ctx.device))
            ctx.profile["families"].setdefault("correlation", []).append((f0, f1))


class Ctx:
    """Everything one forward call needs to know besides its tensors.  Immutable; `replace` derives a variant.

    precision / corr_precision   see PRECISIONS / CORR_PRECISIONS above
    device                       the ROCm device every tensor of the call lives on; kernels are launched on torch's current
                                 stream OF THAT DEVICE, and every pointer handed to the C-ABI is checked against it
    knobs (all default True; parity ablations switch them off one at a time, tools/ablate_parity.py; "16-bit" = the
    single-product modes "bf16" / "f16"):
      use_slab        slab-resident conv / linear kernel (conv_slab16.hip) instead of igemm_bf16.hip (bf16 / bf16x3 only)
      bf16_storage    16-bit: tensors that ONLY feed the next GEMM / the attention kernel live in HBM as 16-bit
      x1_bf16         16-bit: the ResBlock's conv2 output (read by the gate statistics and the apply pass) is 16-bit
      fuse_mlp        16-bit: LayerNorm -> fc1 -> GELU -> fc2 -> +x in one kernel (mlp_fused16.hip)
      fuse_attn       16-bit: LayerNorm -> q/kv GEMMs -> window attention -> proj -> +x in one kernel (attn_fused16.hip)
      commute_upconv  16-bit: relu(conv1x1(bicubic_up(x))) evaluated as relu(bicubic_up(conv1x1(x)))
      commute_any     (default OFF) the same in every arithmetic mode: set by stage overrides that run single layers of a 16-bit
                      frame in split arithmetic (the f32-grade MODES keep the reference's order of operations)
      fuse_apply      (default OFF) 16-bit: the gated residual sum of a ResBlock (all but the last of a stack) is computed inside the next
                      block's first conv while that stages its input (spei_conv_slab16_fa) instead of by spei_resblock_apply.
                      Bit-identical frames; measured no faster: 30.3 vs 30.15 ms per 720p frame
      split_decode    "f16": decoder_second (3 ResBlocks at H/4, 128 channels) and the 1x1 / 3x3 glue convolutions of `_decode`
                      (conv_lv*, search*: model/speinet.py:92-119) run in split arithmetic (bf16x3, f32-grade).  They are 2.4 % of the
                      frame's FLOPs and where the half-operand error weighs most on the `_forwardb` branch: |dPSNR| against the
                      reference on that branch 0.9-1.07e-3 dB -> 1.4-4.6e-4 (every golden <= 4.6e-4; profiles/r03_parity_ablation.txt),
                      for 1.35 ms of a 30.3 ms frame.  Off: round 2's arithmetic (tests then hold 1e-3 dB with no margin on that branch)
      batch_enc       16-bit: the frame's 7 encoder passes (6 without a sharp reference) go through every layer of the three encoder
                      stages in ONE launch per layer (gridDim.y = pass; engine.enc_batched) instead of one launch per pass and layer:
                      bit-identical frames, ~1000 fewer launches per frame, and at H/4 a launch has 3150 workgroups instead of 450
                      (three resident rounds instead of half of one).  Off: round 2's per-pass launches on two streams
      conv3_pipe      16-bit: the 256 -> 256 channel 3x3 convolutions of the Swin body (RSTB tail, conv_after_body) on fp32 token maps whose
                      height is a multiple of 6 and width a multiple of 16 as a persistent pipelined kernel (spei_conv3x3_256_pipe16)
      attn_win4       16-bit, with fuse_attn: the fused attention branch with four windows per workgroup and a batch of maps per launch
                      (spei_attn_win4_16).  Off: round 2's two-window kernel, one map per launch (spei_attn_fused16)
      conv32_ws       16-bit: the 32 -> 32 channel 5x5 convolutions (the ResBlock convs at full resolution: inBlock, outBlock) on the
                      weight-stationary persistent kernel (spei_conv32_ws16: the layer's 51 KB of weights live in each wave's registers, nothing
                      streams from L2 in the main loop).  Off: the slab kernel, which runs these layers at its weight intake
      corr_bf16       "f16" with corr "top2": the candidate pass of the correlation runs on bf16 operands (True, default) instead of
                      f16.  The fp32 re-score decides the winner and S either way (G14: 16 more of 57600 positions differ, dPSNR
                      +1e-6 dB); bf16 operands let the chip hold a ~7 % higher MFMA clock on the slab kernel (tools/bench_corr.py); on the diagonal
                      kernel, which is not limited by the matrix pipe, the difference is 1-5 % (2.60-2.73 vs 2.73-2.76 ms)
      corr_diag       corr "top2", reference map at least as high as the query map (SearchTransfer's maps of one size, SelfTransfer's
                      rotated landscape map): the candidate pass is the diagonal-sliding kernel (corr_diag16.hip): each row-against-row term of the 3x3-patch score is computed once and
                      shared by the three patch rows that use it — a third of the bmm's flops, same fp32 sums.  Off: the slab kernel
    stage        {stage name: {field: value}} overrides applied by `for_stage` (engine: "enc", "swin", "search", "decode", and inside
                 "decode" the stacks "dec2" (decoder_second), "dec1" (decoder_first), "out" (outBlock) and its final conv "tail")
    profile      None, or {op name: [(start_event, end_event), ...]} filled on the launch stream (bench.py)
    capture      None, or a dict that receives intermediate device tensors by name ("arg", "s": what SearchTransfer
                 decided) for the parity tests; eager launches only
    """
    ACT_NONE, ACT_RELU, ACT_GELU = ACT_NONE, ACT_RELU, ACT_GELU
    CONV, CONV_T = CONV, CONV_T
    _FIELDS = ("precision", "corr_precision", "device", "use_slab", "bf16_storage", "x1_bf16", "fuse_mlp", "fuse_attn",
               "commute_upconv", "commute_any", "corr_bf16", "corr_diag", "fuse_apply", "split_decode", "batch_enc", "attn_win4", "conv32_ws", "conv3_pipe", "stage", "profile", "capture")
    # stages of an f16 frame that run in split (bf16x3) arithmetic by default, see `split_decode`
    SPLIT_STAGES = ("glue", "dec2")
    __slots__ = _FIELDS

    def __init__(self, precision: str = "f32", corr_precision: str = "bf16x3", device=None, use_slab: bool = True,
                 bf16_storage: bool = True, x1_bf16: bool = True, fuse_mlp: bool = True, fuse_attn: bool = True,
                 commute_upconv: bool = True, commute_any: bool = False, corr_bf16: bool = True,
                 corr_diag: bool = True, fuse_apply: bool = False, split_decode: bool = True, batch_enc: bool = True,
                 attn_win4: bool = True, conv32_ws: bool = True, conv3_pipe: bool = True, stage: Optional[dict] = None,
                 profile: Optional[dict] = None, capture: Optional[dict] = None):
        if precision not in PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}")
        corr_precision = _CORR_ALIASES.get(corr_precision, corr_precision)
        if corr_precision not in CORR_PRECISIONS:
            raise ValueError(f"unknown correlation precision {corr_precision!r}")
        cur = torch.cuda.current_device() if torch.cuda.is_available() else 0
        device = torch.device("cuda", cur) if device is None else torch.device(device)
        if precision == "f16" and not use_slab:
            raise ValueError("the f16 mode is built on the slab kernels only")
        if device.type != "cuda":
            raise RuntimeError("speinet_amd runs on MI355X only (HIP kernels); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", cur)
        object.__setattr__(self, "precision", precision)
        object.__setattr__(self, "corr_precision", corr_precision)
        object.__setattr__(self, "device", device)
        for k, v in (("use_slab", use_slab), ("bf16_storage", bf16_storage), ("x1_bf16", x1_bf16), ("fuse_mlp", fuse_mlp),
                     ("fuse_attn", fuse_attn), ("commute_upconv", commute_upconv), ("commute_any", commute_any),
                     ("corr_bf16", corr_bf16), ("corr_diag", corr_diag), ("fuse_apply", fuse_apply), ("split_decode", split_decode), ("batch_enc", batch_enc), ("attn_win4", attn_win4), ("conv32_ws", conv32_ws), ("conv3_pipe", conv3_pipe)):
            object.__setattr__(self, k, bool(v))
        object.__setattr__(self, "stage", dict(stage) if stage else {})
        object.__setattr__(self, "profile", profile)
        object.__setattr__(self, "capture", capture)

    def __setattr__(self, k, v):
        raise AttributeError("Ctx is immutable; use replace()")

    def replace(self, **kw) -> "Ctx":
        d = {k: getattr(self, k) for k in self._FIELDS}
        d.update(kw)
        return Ctx(**d)

    def for_stage(self, name: str) -> "Ctx":
        o = self.stage.get(name)
        if o is None and self.split_decode and self.precision == "f16" and name in self.SPLIT_STAGES:
            o = {"precision": "bf16x3", "commute_any": True}
        return self.replace(**o) if o else self

    # ---- plumbing ----------------------------------------------------------------------------------------------
    def _tp(self, t: Optional[torch.Tensor]) -> C.c_void_p:
        if t is None:
            return C.c_void_p(0)
        assert t.is_cuda and t.is_contiguous()
        assert t.device == self.device, f"tensor on {t.device}, call context on {self.device}"
        return C.c_void_p(t.data_ptr())

    def _fp(self, f: Optional[FMap]) -> C.c_void_p:
        if f is None:
            return C.c_void_p(0)
        assert f.t.device == self.device, f"feature map on {f.t.device}, call context on {self.device}"
        return C.c_void_p(f.ptr)

    def _stream(self) -> C.c_void_p:
        # kernels run on the CURRENT HIP device: the caller (SPEINet.forward, detector, tests) holds
        # `torch.cuda.device(ctx.device)`; checked here so a foreign-device launch fails loudly instead of faulting
        assert torch.cuda.current_device() == self.device.index, \
            f"current device cuda:{torch.cuda.current_device()} != call context {self.device}"
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def lp16(self) -> bool:
        """A single-product 16-bit mode ("bf16" or "f16")."""
        return self.precision in ("bf16", "f16")

    @property
    def fmt(self) -> int:
        """16-bit operand format of the matrix-pipe kernels (SPEI_BF16 / SPEI_F16)."""
        return F16 if self.precision == "f16" else BF16

    def inter_dtype(self) -> torch.dtype:
        """Storage type of GEMM-only intermediates (x-hat, q, kv, attention output, MLP hidden, ResBlock conv1 output)."""
        return LP_DTYPE[self.fmt] if (self.lp16 and self.use_slab and self.bf16_storage) else torch.float32

    # ---- K15 / K1 / first and last conv ------------------------------------------------------------------------
    def any_nonzero(self, x: torch.Tensor, flag: torch.Tensor) -> None:
        _lib.check(_lib.lib().spei_any_nonzero(self._tp(x), x.numel(), self._tp(flag), self._stream()), "spei_any_nonzero")

    def rl_prior(self, img: torch.Tensor, iters: int, lam: float = 0.01) -> torch.Tensor:
        """img [3,H,W] -> [3,H,W]."""
        c, h, w = img.shape
        out = torch.empty_like(img)
        scratch = torch.empty_like(img)
        _lib.check(_lib.lib().spei_rl_prior(self._tp(img), self._tp(out), self._tp(scratch), c, h, w, iters, lam, self._stream()),
                   "spei_rl_prior")
        return out

    @_family("conv")
    def conv5_in(self, img: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: Optional[FMap] = None) -> FMap:
        c, h, wd = img.shape
        assert c == 3
        if out is None:
            out = FMap.empty(h, wd, b.numel(), img.device)
        assert (out.H, out.W, out.C, out.ld, out.off) == (h, wd, b.numel(), b.numel(), 0) and not out.lp
        _lib.check(_lib.lib().spei_conv5_in(self._tp(img), self._tp(w), self._tp(b), self._fp(out), h, wd, b.numel(), self._stream()),
                   "spei_conv5_in")
        return out

    @_family("conv")
    def conv5_out(self, f: FMap, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, w32=None,
                  b32: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Last conv, NHWC 32 channels -> three NCHW planes.  w32 / b32 (weights zero-padded to 32 output channels,
        packed): the 16-bit modes run the layer on the slab kernel."""
        assert out.shape == (3, f.H, f.W) and out.is_contiguous() and out.dtype == torch.float32
        if self.lp16 and self.use_slab and w32 is not None and f.C == 32:
            _lib.check(_lib.lib().spei_conv5_out_slab16(self.fmt, self._fp(f), f.ld, f.fmt, self._tp(w32.frag(self.fmt)), self._tp(b32),
                                                       self._tp(out), f.H, f.W, self._stream()), "spei_conv5_out_slab16")
            return out
        if f.lp:                                   # the fp32 kernel reads fp32 maps (only reached through a stage override)
            f = FMap(f.t.float(), f.H, f.W, f.C, f.off)
        _lib.check(_lib.lib().spei_conv5_out(self._fp(f), f.ld, self._tp(w), self._tp(b), self._tp(out), f.H, f.W, f.C, self._stream()),
                   "spei_conv5_out")
        return out

    # ---- the GEMM family ---------------------------------------------------------------------------------------
    def _conv3_pipe_ok(self, a, w, N: int, ksize: int, stride: int, act: int, residual, out) -> bool:
        """The persistent 3x3 / 256-channel kernel takes the call: dense fp32 maps, 6 x 16 pixel tiles cover the map, 16-bit single products."""
        return (self.conv3_pipe and self.lp16 and self.use_slab and ksize == 3 and stride == 1 and N == 256 and a.C == 256 and act == ACT_NONE
                and a.H % 6 == 0 and a.W % 16 == 0 and a.H * a.W <= (1 << 21) and a.t.dtype == torch.float32 and getattr(a, "ld", 256) == 256
                and getattr(a, "off", 0) == 0 and getattr(w, "fhi", None) is not None and tuple(w.shape) == (9, 256, 256)
                and (out is None or (out.t.dtype == torch.float32 and getattr(out, "ld", 256) == 256 and getattr(out, "off", 0) == 0
                                     and out.t.data_ptr() != a.t.data_ptr()))
                and (residual is None or (residual.t.dtype == torch.float32 and getattr(residual, "ld", 256) == 256 and getattr(residual, "off", 0) == 0)))

    @_family("conv")
    def igemm(self, a0: FMap, w, bias: Optional[torch.Tensor], N: int, ksize: int = 1, stride: int = 1,
              mode: int = CONV, act: int = ACT_NONE, a1: Optional[FMap] = None, residual: Optional[FMap] = None,
              rowscale: Optional[torch.Tensor] = None, out: Optional[FMap] = None, out_dtype=torch.float32,
              ln_input: bool = False) -> FMap:
        prec = self.precision
        pad = ksize // 2
        if mode == CONV:
            ho, wo = (a0.H + 2 * pad - ksize) // stride + 1, (a0.W + 2 * pad - ksize) // stride + 1
        else:
            ho, wo = a0.H * stride, a0.W * stride
        if out is None:
            out = FMap.empty(ho, wo, N, a0.t.device, out_dtype)
        assert out.H == ho and out.W == wo and out.C == N
        k0, k1 = a0.C, (a1.C if a1 is not None else 0)
        slab = prec != "f32" and self.use_slab and mode == CONV
        assert slab or (mode == CONV_T and self.lp16 and self.use_slab) or not (a0.lp or out.lp), \
            "16-bit activations are only supported by the slab kernel"
        assert a1 is None or a1.t.dtype == a0.t.dtype
        assert residual is None or not residual.lp
        assert all(f.t.dtype in (torch.float32, LP_DTYPE[self.fmt]) for f in (a0, out)), "16-bit tensors must be in the mode's format"
        if torch.is_tensor(w):
            w = PackedW(w, a0.t.device)
        assert not ln_input or (slab and w.fhi is not None), "ln_input is a feature of the slab kernel"
        assert tuple(w.shape) == (ksize * ksize, N, k0 + k1), (tuple(w.shape), ksize, N, k0, k1)
        if a1 is not None:
            assert (a1.H, a1.W) == (a0.H, a0.W)
        if residual is not None:
            assert (residual.H, residual.W, residual.C) == (ho, wo, N)
        if rowscale is not None:
            assert rowscale.numel() == ho * wo
        tp, fp = self._tp, self._fp
        common = (fp(out), out.ld, fp(residual), residual.ld if residual is not None else 0, tp(rowscale), a0.H, a0.W, ho, wo, N,
                  ksize, stride, pad, mode, act, self._stream())
        srcs = (fp(a0), a0.ld, k0, fp(a1), a1.ld if a1 is not None else 0, k1)
        lib = _lib.lib()
        if (mode == CONV and a1 is None and rowscale is None and not ln_input and out_dtype == torch.float32
                and self._conv3_pipe_ok(a0, w, N, ksize, stride, act, residual, out)):
            _lib.check(lib.spei_conv3x3_256_pipe16(self.fmt, fp(a0), tp(w.frag(self.fmt)), tp(bias), fp(residual), fp(out), 1, a0.H, a0.W,
                                                   self._stream()), "spei_conv3x3_256_pipe16")
        elif (self.conv32_ws_available() and mode == CONV and ksize == 5 and stride == 1 and k0 == 32 and N == 32 and a1 is None
                and residual is None and rowscale is None and not ln_input and act in (ACT_NONE, ACT_RELU) and w.fhi is not None
                and (a0.ld, a0.off, out.ld, out.off) == (32, 0, 32, 0) and a0.t.data_ptr() != out.t.data_ptr()):
            # the 32-channel 5x5 layers: weight-stationary persistent kernel (csrc/conv32_ws16.hip)
            _lib.check(lib.spei_conv32_ws16(self.fmt, fp(a0), a0.fmt, tp(w.frag(self.fmt)), tp(bias), fp(out), out.fmt, 1, a0.H, a0.W, act,
                                            self._stream()), "spei_conv32_ws16")
        elif (mode == CONV_T and self.lp16 and self.use_slab and ksize == 3 and stride == 2 and a1 is None and residual is None
                and rowscale is None and N % 32 == 0 and k0 % 32 == 0):
            # stride-2 transposed conv = four stride-1 convs (one per output parity) on the slab kernel
            cf = w.convT_class_frags(self.fmt)
            _lib.check(lib.spei_convt2_slab16(self.fmt, fp(a0), a0.ld, k0, a0.fmt, tp(cf[(0, 0)]), tp(cf[(0, 1)]), tp(cf[(1, 0)]),
                                              tp(cf[(1, 1)]), tp(bias), fp(out), out.ld, out.fmt, a0.H, a0.W, N, act,
                                              self._stream()), "spei_convt2_slab16")
        elif (mode == CONV_T and prec == "bf16x3" and self.use_slab and ksize == 3 and stride == 2 and a1 is None and residual is None
                and rowscale is None and N % 32 == 0 and k0 % 32 == 0 and not a0.lp and not out.lp):
            # the same in split arithmetic (the transposed conv that ends decoder_second inside an f16 frame's split stages; round 1's igemm
            # kernel took 327 us for it)
            _hi, _lo, ph, pl = w.convT_class_frags_split()
            _lib.check(lib.spei_convt2_slab16x3(fp(a0), a0.ld, k0, ph, pl, tp(bias), fp(out), out.ld, a0.H, a0.W, N, act, self._stream()),
                       "spei_convt2_slab16x3")
        elif prec == "f32":
            _lib.check(lib.spei_igemm_f32(*srcs, tp(w.f32), tp(bias), *common), "spei_igemm_f32")
        elif slab and w.fhi is not None:
            dims = (a0.H * a0.W, 1, ho * wo, 1) if (ksize == 1 and stride == 1) else (a0.H, a0.W, ho, wo)
            _lib.check(lib.spei_conv_slab16(
                self.fmt, *srcs, a0.fmt, tp(w.frag(self.fmt)), tp(w.flo) if prec == "bf16x3" else _vp(0), tp(bias), fp(out), out.ld,
                out.fmt, fp(residual), residual.ld if residual is not None else 0,
                tp(rowscale), *dims, N, ksize, stride, pad, act, int(ln_input), self._stream()), "spei_conv_slab16")
        else:
            assert prec != "f16", "f16: N and K must be multiples of 32 (slab kernel)"
            _lib.check(lib.spei_igemm_bf16(*srcs, tp(w.hi), tp(w.lo) if prec == "bf16x3" else _vp(0), tp(bias), *common),
                       "spei_igemm_bf16")
        return out

    def linear(self, x: torch.Tensor, w, b: Optional[torch.Tensor], act: int = ACT_NONE,
               residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_dtype=torch.float32,
               ln_input: bool = False) -> torch.Tensor:
        """Token-space linear: x [M,K] -> [M,N]; w [N,K]."""
        m, k = x.shape
        if torch.is_tensor(w):
            w = PackedW(w.reshape(1, *w.shape), x.device)
        n = w.shape[1]
        if out is None:
            out = torch.empty(m, n, device=x.device, dtype=out_dtype)
        self.igemm(FMap(x, m, 1, k), w, b, n, act=act, ln_input=ln_input,
                   residual=FMap(residual, m, 1, n) if residual is not None else None, out=FMap(out, m, 1, n))
        return out

    # ---- ResBlock (K3) -----------------------------------------------------------------------------------------
    @_family("streaming")
    def resblock_gates(self, x1: FMap, pk: dict):
        dev = x1.t.device
        lib = _lib.lib()
        tp = self._tp
        assert x1.off == 0 and x1.ld == x1.C
        s = torch.empty(x1.C, device=dev)
        g1 = torch.empty(x1.H, x1.C, device=dev)
        g2 = torch.empty(x1.W, x1.C, device=dev)
        ws = torch.empty(lib.spei_gate_ws_floats(x1.H, x1.W, x1.C), device=dev)
        _lib.check(lib.spei_resblock_gates(self._fp(x1), x1.fmt, x1.H, x1.W, x1.C, tp(pk["se_w1"]), tp(pk["se_b1"]), tp(pk["se_w2"]),
                                           tp(pk["se_b2"]), tp(pk["cw_w"]), tp(pk["cw_bn"]), tp(pk["hc_w"]), tp(pk["hc_bn"]),
                                           tp(s), tp(g1), tp(g2), tp(ws), self._stream()), "spei_resblock_gates")
        return s, g1, g2

    def apply_fused_available(self, c: int) -> bool:
        """The gated residual sum of a ResBlock can ride in the NEXT block's first conv (spei_conv_slab16_fa): 16-bit modes on the
        slab kernel with a 16-bit x1."""
        return self.lp16 and self.use_slab and self.x1_bf16 and self.fuse_apply and c in (32, 64, 128, 256)

    def resblock(self, x, pk: dict, extra: Optional[FMap] = None, out: Optional[FMap] = None, defer: bool = False):
        """x + SE(x1) + TE(x1), x1 = conv5(relu(conv5(x)))  (reference model/block.py:127-140).

        `x` is a map, or the deferred tail of the previous block — a tuple (x, x1, s, g1, g2): then this block's first conv computes
        that block's output while it stages its input (one kernel fewer, one fp32 map round trip fewer) and writes it back as the
        residual stream.  `defer=True` returns this block's own tail instead of applying it (the caller passes it to the next)."""
        idt = self.inter_dtype()
        if isinstance(x, tuple):
            px, px1, ps, pg1, pg2 = x
            c = px.C
            xn = FMap.empty(px.H, px.W, c, px.t.device)
            t = FMap.empty(px.H, px.W, c, px.t.device, idt)
            w1 = pk["w1"] if not torch.is_tensor(pk["w1"]) else PackedW(pk["w1"], px.t.device)
            _lib.check(_lib.lib().spei_conv_slab16_fa(self.fmt, self._fp(px), c, self._fp(px1), self._tp(ps), self._tp(pg1), self._tp(pg2),
                                                      self._fp(xn), self._tp(w1.frag(self.fmt)), self._tp(pk["b1"]), self._fp(t), t.ld, t.fmt,
                                                      px.H, px.W, c, 5, ACT_RELU, self._stream()), "spei_conv_slab16_fa")
            x = xn
        else:
            c = x.C
            assert x.off == 0 and x.ld == c
            t = self.igemm(x, pk["w1"], pk["b1"], c, ksize=5, act=ACT_RELU, out_dtype=idt)   # only conv2 reads it
        x1 = self.igemm(t, pk["w2"], pk["b2"], c, ksize=5, out_dtype=idt if self.x1_bf16 else torch.float32)
        s, g1, g2 = self.resblock_gates(x1, pk)
        if defer:
            return (x, x1, s, g1, g2)
        if out is None:
            out = FMap.empty(x.H, x.W, c, x.t.device)
        if extra is not None:
            assert extra.off == 0 and extra.ld == c
        self._apply(x, x1, s, g1, g2, extra, out, c)
        return out

    @_family("streaming")
    def _apply(self, x, x1, s, g1, g2, extra, out, c):
        _lib.check(_lib.lib().spei_resblock_apply(self._fp(x), self._fp(x1), x1.fmt, self._tp(s), self._tp(g1), self._tp(g2),
                                                  self._fp(extra), self._fp(out), out.ld, x.H, x.W, c, self._stream()),
                   "spei_resblock_apply")

    # ---- the same stacks on several maps per launch (the frame's encoder passes) -------------------------------------------------
    def batched_available(self) -> bool:
        """One launch per layer for all maps of a BMap (the frame's encoder passes): the single-product 16-bit modes on the slab kernel."""
        return self.batched_kernels() and self.x1_bf16 and self.bf16_storage and not self.fuse_apply and self.batch_enc

    def conv32_ws_available(self) -> bool:
        """The weight-stationary kernel for 32 -> 32 channel 5x5 layers (single-product 16-bit modes)."""
        return self.lp16 and self.use_slab and self.conv32_ws

    def batched_kernels(self) -> bool:
        """`igemm_batched` can run (what the batched Swin calls need; `batch_enc` only decides about the encoder passes)."""
        return self.lp16 and self.use_slab

    @_family("conv")
    def igemm_batched(self, a: BMap, w, bias: torch.Tensor, N: int, ksize: int, stride: int = 1, act: int = ACT_NONE,
                      out_dtype=torch.float32, residual: Optional[BMap] = None, out: Optional[BMap] = None) -> BMap:
        """One conv layer on every map of `a` in one launch (gridDim.y = map).  residual: fp32 maps of the output shape added in the
        epilogue; it may be `out` itself (each workgroup reads the residual of its own pixels before it writes them), never `a`."""
        assert self.batched_kernels() and not torch.is_tensor(w) and w.fhi is not None
        pad = ksize // 2
        ho, wo = (a.H + 2 * pad - ksize) // stride + 1, (a.W + 2 * pad - ksize) // stride + 1
        assert tuple(w.shape) == (ksize * ksize, N, a.C) and a.t.dtype in (torch.float32, LP_DTYPE[self.fmt])
        if out is None:
            out = BMap.empty(a.B, ho, wo, N, a.t.device, out_dtype)
        assert (out.B, out.H, out.W, out.C) == (a.B, ho, wo, N) and out.t.data_ptr() != a.t.data_ptr()
        if residual is not None:
            assert (residual.B, residual.H, residual.W, residual.C) == (a.B, ho, wo, N) and residual.t.dtype == torch.float32
            assert residual.t.data_ptr() != a.t.data_ptr()
        tp = self._tp
        if out_dtype == torch.float32 and self._conv3_pipe_ok(a, w, N, ksize, stride, act, residual, out):
            _lib.check(_lib.lib().spei_conv3x3_256_pipe16(self.fmt, tp(a.t), tp(w.frag(self.fmt)), tp(bias),
                                                          tp(residual.t) if residual is not None else _vp(0), tp(out.t), a.B, a.H, a.W,
                                                          self._stream()), "spei_conv3x3_256_pipe16")
            return out
        if (self.conv32_ws_available() and ksize == 5 and stride == 1 and a.C == 32 and N == 32 and residual is None
                and act in (ACT_NONE, ACT_RELU)):
            _lib.check(_lib.lib().spei_conv32_ws16(self.fmt, tp(a.t), a.fmt, tp(w.frag(self.fmt)), tp(bias), tp(out.t), out.fmt, a.B, a.H, a.W,
                                                   act, self._stream()), "spei_conv32_ws16")
            return out
        _lib.check(_lib.lib().spei_conv_slab16_batched(self.fmt, tp(a.t), a.C, a.fmt, tp(w.frag(self.fmt)), _vp(0), tp(bias), tp(out.t), out.fmt,
                                                       tp(residual.t) if residual is not None else _vp(0), a.B, a.H, a.W, ho, wo, N, ksize,
                                                       stride, pad, act, self._stream()),
                   "spei_conv_slab16_batched")
        return out

    def resblock_batched(self, x: BMap, pk: dict) -> BMap:
        """`resblock` on every map of x: conv1, conv2, gate statistics, gate maps and the gated sum are ONE launch each."""
        idt = self.inter_dtype()
        c, dev = x.C, x.t.device
        lib = _lib.lib()
        tp = self._tp
        t = self.igemm_batched(x, pk["w1"], pk["b1"], c, 5, act=ACT_RELU, out_dtype=idt)
        x1 = self.igemm_batched(t, pk["w2"], pk["b2"], c, 5, out_dtype=idt)
        del t
        return self._gates_apply_batched(x, x1, pk)

    @_family("streaming")
    def _gates_apply_batched(self, x: BMap, x1: BMap, pk: dict) -> BMap:
        c, dev = x.C, x.t.device
        lib = _lib.lib()
        tp = self._tp
        s = torch.empty(x.B, c, device=dev)
        g1 = torch.empty(x.B, x.H, c, device=dev)
        g2 = torch.empty(x.B, x.W, c, device=dev)
        ws = torch.empty(x.B * lib.spei_gate_ws_floats(x.H, x.W, c), device=dev)
        _lib.check(lib.spei_resblock_gates_batched(tp(x1.t), x1.fmt, x.B, x.H, x.W, c, tp(pk["se_w1"]), tp(pk["se_b1"]), tp(pk["se_w2"]),
                                                   tp(pk["se_b2"]), tp(pk["cw_w"]), tp(pk["cw_bn"]), tp(pk["hc_w"]), tp(pk["hc_bn"]),
                                                   tp(s), tp(g1), tp(g2), tp(ws), self._stream()), "spei_resblock_gates_batched")
        out = BMap.empty(x.B, x.H, x.W, c, dev)
        _lib.check(lib.spei_resblock_apply_batched(tp(x.t), tp(x1.t), x1.fmt, tp(s), tp(g1), tp(g2), tp(out.t), x.B, x.H, x.W, c,
                                                   self._stream()), "spei_resblock_apply_batched")
        return out

    # ---- Swin (K6-K9) ------------------------------------------------------------------------------------------
    def ln_fused_available(self) -> bool:
        """LayerNorm folded into the staging of the following 256-wide linear (slab kernel, every mode but f32)."""
        return self.precision != "f32" and self.use_slab

    def attn_fused_available(self) -> bool:
        return self.lp16 and self.use_slab and self.fuse_attn and self.bf16_storage

    def mlp_fused_available(self) -> bool:
        return self.lp16 and self.use_slab and self.fuse_mlp

    def swin_multi_available(self) -> bool:
        """The Swin calls of a frame as one batch over stacked maps (engine.swin_multi): needs the batched attention kernel."""
        return self.attn_fused_available() and self.attn_win4 and self.mlp_fused_available() and self.batched_kernels()

    @_family("swin")
    def attn_fused(self, x: torch.Tensor, yhat: torch.Tensor, bk: dict, H: int, W: int, shift: int, out: torch.Tensor) -> torch.Tensor:
        """out = x + proj(window_attention(...))  (reference model/swinir.py:238-278); in place when out is x.  x, yhat, out:
        [B * H * W, 256], B >= 1 equally sized maps stacked (the Swin calls of a frame share the block's weights); B > 1 needs the
        batched four-window kernel (`attn_win4`)."""
        assert x.shape[0] % (H * W) == 0 and x.shape[1] == 256 and x.dtype == torch.float32 and out.shape == x.shape and out.dtype == torch.float32
        f = self.fmt
        assert yhat.shape == x.shape and yhat.dtype == LP_DTYPE[f]
        tp = self._tp
        batch = x.shape[0] // (H * W)
        if self.attn_win4:
            _lib.check(_lib.lib().spei_attn_win4_16(f, tp(x), tp(out), tp(yhat), tp(bk["wq"].frag(f)), tp(bk["bq"]), tp(bk["wkv"].frag(f)),
                                                    tp(bk["bkv"]), tp(bk["wproj"].frag(f)), tp(bk["bproj"]), tp(bk["relbias"]), batch, H, W,
                                                    shift, self._stream()), "spei_attn_win4_16")
            return out
        assert batch == 1, "the two-window kernel takes one map per launch"
        _lib.check(_lib.lib().spei_attn_fused16(f, tp(x), tp(out), tp(yhat), tp(bk["wq"].frag(f)), tp(bk["bq"]), tp(bk["wkv"].frag(f)),
                                                tp(bk["bkv"]), tp(bk["wproj"].frag(f)), tp(bk["bproj"]), tp(bk["relbias"]), H, W, shift,
                                                self._stream()), "spei_attn_fused16")
        return out

    @_family("swin")
    def mlp_fused(self, x: torch.Tensor, w1, b1: torch.Tensor, w2, b2: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """out = x + fc2(gelu(fc1(LN(x))))  (reference model/swinir.py:279); in place when out is x."""
        assert x.shape[1] == 256 and x.dtype == torch.float32 and out.shape == x.shape and out.dtype == torch.float32
        assert tuple(w1.shape) == (1, 512, 256) and tuple(w2.shape) == (1, 256, 512)
        tp = self._tp
        f = self.fmt
        _lib.check(_lib.lib().spei_mlp_fused16(f, tp(x), tp(out), tp(w1.frag(f)), tp(b1), tp(w2.frag(f)), tp(b2), x.shape[0], self._stream()),
                   "spei_mlp_fused16")
        return out

    def layernorm(self, x: torch.Tensor, g: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, out_dtype=torch.float32) -> torch.Tensor:
        assert x.shape[1] == 256 and x.is_contiguous() and x.dtype == torch.float32
        if out is None:
            out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
        tp = self._tp
        _lib.check(_lib.lib().spei_layernorm256(tp(x), tp(out), fmt_of(out.dtype), tp(g), tp(b), x.shape[0], self._stream()),
                   "spei_layernorm256")
        return out

    def window_attention(self, q: torch.Tensor, kv: torch.Tensor, relbias: torch.Tensor, H: int, W: int, shift: int,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert q.shape == (H * W, 256) and kv.shape == (H * W, 512) and relbias.shape == (8, 25, 25)
        if out is None:
            out = torch.empty_like(q)
        assert q.dtype == kv.dtype == out.dtype
        tp = self._tp
        _lib.check(_lib.lib().spei_window_attention(tp(q), tp(kv), fmt_of(q.dtype), tp(relbias), tp(out), H, W, shift,
                                                    self._stream()), "spei_window_attention")
        return out

    # ---- SearchTransfer (K10-K12) ------------------------------------------------------------------------------
    def patch_invnorm(self, f: FMap) -> torch.Tensor:
        inv = torch.empty(f.H * f.W, device=f.t.device)
        _lib.check(_lib.lib().spei_patch_invnorm(self._fp(f), f.ld, self._tp(inv), f.H, f.W, f.C, self._stream()), "spei_patch_invnorm")
        return inv

    def corr_plan(self, lr: FMap, ref: FMap, inv_lr: torch.Tensor, inv_ref: torch.Tensor) -> "CorrPlan":
        """Everything of K11 except the arg-max kernel itself: output / workspace allocation and the one-off 16-bit
        conversion of the two maps.  `plan.launch()` then runs the dominant kernel of the path (timed by `profile`), so a
        caller can place it between two captured graph segments and bracket it with HIP events (bench.py)."""
        lib = _lib.lib()
        dev = lr.t.device
        tp, fp = self._tp, self._fp
        n = lr.H * lr.W
        s = torch.empty(n, device=dev)
        arg = torch.empty(n, device=dev, dtype=torch.int32)
        dims = (lr.H, lr.W, ref.H, ref.W, lr.C)
        rescore = self.precision != "f32" and self.corr_precision == "top2" and self.use_slab and lr.C == 128
        # the diagonal kernel keeps one candidate pair per (query, diagonal, reference tile): 415 MB at 720p, ~34 GB at 4K — beyond the
        # budget (a quarter of the device's free memory, at most 8 GiB) the slab kernel takes over (its workspace is per query only)
        diag, diag_floats = False, 0
        if rescore and self.corr_diag and ref.H >= lr.H:
            diag_floats = lib.spei_corr_diag_ws_floats(lr.H, lr.W, ref.H, ref.W)
            diag = 4 * diag_floats <= min(torch.cuda.mem_get_info(dev)[0] // 4, CORR_DIAG_WS_MAX)
        ws = torch.empty(diag_floats if diag else lib.spei_corr_ws_floats(n), device=dev)
        if self.precision == "f32":
            args = (fp(lr), lr.ld, fp(ref), ref.ld, tp(inv_lr), tp(inv_ref), *dims, tp(s), tp(arg), tp(ws))
            return CorrPlan(self, s, arg, "corr_argmax_kernel (f32 MFMA)", [(lib.spei_corr_argmax, "spei_corr_argmax", args)], [],
                            (lr, ref, inv_lr, inv_ref, ws))
        split = self.corr_precision == "bf16x3"
        f16 = BF16 if (rescore and self.corr_bf16) else self.fmt
        assert f16 == BF16 or (not split and self.use_slab and lr.C == 128), "f16 correlation: slab kernel, single / top2"
        parts = []
        for f in (lr, ref):
            hi = torch.empty(f.H * f.W, f.C, device=dev, dtype=LP_DTYPE[f16])
            lo = torch.empty_like(hi) if split else None
            _lib.check(lib.spei_split16(f16, fp(f), f.ld, tp(hi), tp(lo), f.H * f.W, f.C, self._stream()), "spei_split16")
            parts += [hi, lo]
        fname = "f16" if f16 == F16 else "bf16"
        if rescore:
            arg2 = torch.empty(n, device=dev, dtype=torch.int32)
            s2 = torch.empty(n, device=dev)
            if diag:
                main = (lib.spei_corr_diag_top2_16, "spei_corr_diag_top2_16",
                        (f16, tp(parts[0]), tp(parts[2]), tp(inv_ref), *dims, tp(s), tp(arg), tp(s2), tp(arg2), tp(ws)))
            else:
                main = (lib.spei_corr_slab_top2_16, "spei_corr_slab_top2_16",
                        (f16, tp(parts[0]), tp(parts[2]), tp(inv_lr), tp(inv_ref), *dims, tp(s), tp(arg), tp(s2), tp(arg2), tp(ws)))
            post = (lib.spei_corr_rescore, "spei_corr_rescore",
                    (fp(lr), lr.ld, fp(ref), ref.ld, tp(inv_lr), tp(inv_ref), *dims, tp(s), tp(arg), tp(s2), tp(arg2)))
            return CorrPlan(self, s, arg, f"corr_{'diag' if diag else 'slab'}_kernel<top2, {fname}>", [main], [post],
                            (lr, ref, inv_lr, inv_ref, ws, parts, s2, arg2))
        args = (tp(parts[0]), tp(parts[1]), tp(parts[2]), tp(parts[3]), tp(inv_lr), tp(inv_ref), *dims, tp(s), tp(arg), tp(ws))
        if self.use_slab and lr.C == 128:
            main = (lib.spei_corr_slab16, "spei_corr_slab16", (f16, *args))
            kname = f"corr_slab_kernel<{'bf16x3' if split else 'top1, ' + fname}>"
        else:
            main = (lib.spei_corr_argmax_bf16, "spei_corr_argmax_bf16", args)
            kname = f"corr_bf16_kernel<{'bf16x3' if split else 'bf16'}>"
        return CorrPlan(self, s, arg, kname, [main], [], (lr, ref, inv_lr, inv_ref, ws, parts))

    def corr_argmax(self, lr: FMap, ref: FMap, inv_lr: torch.Tensor, inv_ref: torch.Tensor):
        plan = self.corr_plan(lr, ref, inv_lr, inv_ref)
        plan.launch()
        return plan.s, plan.arg

    def gather_fold(self, ref: FMap, arg: torch.Tensor, H3: int, W3: int, Hr3: int, Wr3: int, s: int) -> FMap:
        assert ref.H == Hr3 * s and ref.W == Wr3 * s
        out = FMap.empty(H3 * s, W3 * s, ref.C, ref.t.device)
        _lib.check(_lib.lib().spei_gather_fold(self._fp(ref), ref.ld, self._tp(arg), self._fp(out), out.ld, H3, W3, Hr3, Wr3, ref.C, s,
                                               self._stream()), "spei_gather_fold")
        return out

    # ---- glue (K13-K14) ----------------------------------------------------------------------------------------
    def rot90(self, f: FMap) -> FMap:
        out = FMap.empty(f.W, f.H, f.C, f.t.device)
        _lib.check(_lib.lib().spei_rot90(self._fp(f), f.ld, self._fp(out), f.H, f.W, f.C, self._stream()), "spei_rot90")
        return out

    def upsample(self, f: FMap, s: int, act: int = ACT_NONE) -> FMap:
        out = FMap.empty(f.H * s, f.W * s, f.C, f.t.device)
        _lib.check(_lib.lib().spei_upsample_bicubic(self._fp(f), f.ld, self._fp(out), out.ld, f.H, f.W, f.C, s, act, self._stream()),
                   "spei_upsample_bicubic")
        return out

    def up_conv1x1_relu(self, f: FMap, w, b: torch.Tensor, n: int, s: int = 2) -> FMap:
        """relu(conv1x1(bicubic_up(f))) (reference model/speinet.py:96-97,108-109, model/SearchTransfer.py:73-76).  Both maps
        are linear and the bicubic weights sum to 1, so the 16-bit modes run the conv first, at 1/s^2 of the pixels and with
        half the bytes through the upsampler; the f32-grade modes keep the reference's order of operations."""
        if (self.lp16 and self.commute_upconv) or self.commute_any:
            return self.upsample(self.igemm(f, w, b, n), s, act=ACT_RELU)
        return self.igemm(self.upsample(f, s), w, b, n, act=ACT_RELU)

    def add(self, a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(a)
        assert out.shape == a.shape == b.shape and out.dtype == a.dtype == torch.float32
        _lib.check(_lib.lib().spei_add(self._tp(a), self._tp(b), self._tp(out), a.numel(), self._stream()), "spei_add")
        return out
