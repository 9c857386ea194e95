"""First slice of the training step: `torch.autograd.Function`s whose forward AND backward run on the HIP
kernels, for everything the recons_net stacks are made of — Conv2d (+ fused ReLU) and the ResBlock with its SE / triplet
gates (reference model/recons_video_ori.py:26-84, model/block.py:8-140) — and `encoder_train`, the differentiable
`encoder_second(encoder_first(inBlock(frame)))` built from them on the parameters of the drop-in module itself, so
`loss.backward()` fills `.grad` of the same `nn.Parameter`s the reference's optimizer would step
(trainer/trainer_swint_hsa_nsf.py:14-16,34-40).

Arithmetic: fp32 (`v_mfma_f32_32x32x2_f32`), gradients pinned against the reference's own autograd (tests/golden/make_golden_grad.py).
BatchNorm(1) of the gates uses its running statistics and DropPath is the identity: the eval-mode graph, as the round-2 slice
states in DESIGN.md; batch statistics and DropPath masks are the next step.

What runs where:
  HIP    conv forward (spei_igemm_f32 / spei_conv5_in), conv data gradient (spei_igemm_f32, transposed mode), conv weight + bias
         gradient (spei_conv_wgrad_f32), ReLU mask (spei_relu_bwd), the gates' plane statistics and their backward sums
         (spei_plane_stats), the gated residual sum and its backward (spei_resblock_apply, spei_resblock_apply_bwd)
  torch  the gate maps themselves — an SE MLP on C numbers and two 2->1 channel convolutions on [H][C] / [C][W] planes, a few
         thousand values per ResBlock — are evaluated with torch ops on the device and differentiated by torch.autograd inside
         the Function's backward: plumbing-sized work, no pixel-sized tensor ever goes through torch arithmetic.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib, pack
from .ops import ACT_NONE, ACT_RELU, CONV_T, Ctx, FMap


def _ctx(device) -> Ctx:
    return Ctx("f32", "bf16x3", device=device)


class _Conv(torch.autograd.Function):
    """y[Ho*Wo, N] = act(conv(x[H*W, K], weight[N, K, k, k]) + bias)   (NHWC pixel rows)."""

    @staticmethod
    def forward(fctx, x, weight, bias, H, W, ksize, stride, relu):
        ctx = _ctx(x.device)
        n, k = weight.shape[0], weight.shape[1]
        xf = FMap(x.contiguous(), H, W, k)
        out = ctx.igemm(xf, pack.conv_w(weight.detach()), bias.detach().contiguous(), n, ksize=ksize, stride=stride,
                        act=ACT_RELU if relu else ACT_NONE)
        fctx.save_for_backward(xf.t, weight, out.t)
        fctx.meta = (H, W, out.H, out.W, ksize, stride, relu)
        return out.t

    @staticmethod
    def backward(fctx, dy):
        x, weight, y = fctx.saved_tensors
        H, W, Ho, Wo, ksize, stride, relu = fctx.meta
        return _conv_backward(x, weight, y, dy, H, W, Ho, Wo, ksize, stride, relu, fctx.needs_input_grad[0]) + (None,) * 5


def _conv_backward(x, weight, y, dy, H, W, Ho, Wo, ksize, stride, relu, need_dx):
    ctx = _ctx(dy.device)
    lib = _lib.lib()
    tp = ctx._tp
    n, k = weight.shape[0], weight.shape[1]
    dy = dy.contiguous()
    if relu:
        dz = torch.empty_like(dy)
        _lib.check(lib.spei_relu_bwd(tp(y), tp(dy), tp(dz), dy.numel(), ctx._stream()), "spei_relu_bwd")
        dy = dz
    dw = torch.empty(ksize * ksize, n, k, device=dy.device)
    db = torch.empty(n, device=dy.device)
    ws = torch.empty(lib.spei_wgrad_ws_floats(Ho, Wo, n, k, ksize), device=dy.device)
    _lib.check(lib.spei_conv_wgrad_f32(tp(x), x.shape[1], tp(dy), n, tp(dw), tp(db), tp(ws), H, W, Ho, Wo, n, k, ksize, stride, ksize // 2,
                                       ctx._stream()), "spei_conv_wgrad_f32")
    dweight = dw.view(ksize, ksize, n, k).permute(2, 3, 0, 1).contiguous()          # [t][n][k] -> [N, K, k, k]
    dx = None
    if need_dx:
        # dX = transposed convolution of dY with the weights' channel axes swapped (same taps): include/speinet_hip.h
        wt = pack.conv_w(weight.detach()).transpose(1, 2).contiguous()                  # [t][k][n]
        dx = ctx.igemm(FMap(dy, Ho, Wo, n), wt, None, k, ksize=ksize, stride=stride, mode=CONV_T).t
        if dx.shape[0] != H * W:          # odd input size under stride 2: the transposed conv produced one row / column too many
            dx = dx.view(Ho * stride, Wo * stride, k)[:H, :W].reshape(H * W, k).contiguous()
    return dx, dweight, db


class _ConvIn(torch.autograd.Function):
    """The 3-channel head conv on NCHW planes (spei_conv5_in, 5x5 + ReLU); the frame itself needs no gradient."""

    @staticmethod
    def forward(fctx, frame, weight, bias):
        ctx = _ctx(frame.device)
        c, H, W = frame.shape
        out = ctx.conv5_in(frame.contiguous(), pack.conv_w(weight.detach()).to(frame.device), bias.detach().contiguous())
        fctx.save_for_backward(frame.permute(1, 2, 0).reshape(H * W, c).contiguous(), weight, out.t)
        fctx.meta = (H, W)
        return out.t

    @staticmethod
    def backward(fctx, dy):
        x, weight, y = fctx.saved_tensors
        H, W = fctx.meta
        _, dweight, db = _conv_backward(x, weight, y, dy, H, W, H, W, 5, 1, True, False)
        return None, dweight, db


def _gate_maps(rowmax, rowmean, colmax, colmean, mean, se_w1, se_b1, se_w2, se_b2, cw_w, cw_bn, hc_w, hc_bn):
    """s [C], g1 [H][C], g2 [W][C] from the plane statistics (reference model/block.py:8-24 SE, :75-96 gates without sigmoid,
    :49-68 BasicConv1 with eval BatchNorm(1); TripletAttention sums the two gates, :116-119).  cw_bn / hc_bn: (weight, bias,
    running_mean, running_var)."""
    s = torch.sigmoid(F.linear(F.relu(F.linear(mean, se_w1, se_b1)), se_w2, se_b2))

    def bn(t, p):
        w, b, rm, rv = p
        return (t - rm) / torch.sqrt(rv + 1e-5) * w + b

    def conv21(z, w, k):
        # the 2 -> 1 channel k x k convolution as unfold + one matrix product: the library convolution's weight gradient is not
        # bitwise reproducible for these shapes (atomics), this form is
        a, b = z.shape[-2:]
        return (w.reshape(1, -1) @ F.unfold(z, k, padding=k // 2)[0]).view(1, 1, a, b)

    z1 = torch.stack((rowmax, rowmean)).unsqueeze(0)                       # [1, 2, H, C]: conv "height" = H, "width" = C
    g1 = bn(conv21(z1, cw_w, 7), cw_bn)[0, 0]                              # [H, C]
    z2 = torch.stack((colmax.t(), colmean.t())).unsqueeze(0)               # [1, 2, C, W]: conv "height" = C, "width" = W
    g2 = bn(conv21(z2, hc_w, 5), hc_bn)[0, 0].t()                          # [W, C]
    return s, g1.contiguous(), g2.contiguous()


class _GatedSum(torch.autograd.Function):
    """out = x + x1 * (s + g1[y] + g2[x]) with the gates built from x1's plane statistics (the tail of model/block.py:136-140)."""
    N_PARAM = 14

    @staticmethod
    def forward(fctx, x, x1, H, W, *params):
        ctx = _ctx(x.device)
        lib = _lib.lib()
        tp = ctx._tp
        c = x.shape[1]
        dev = x.device
        x1 = x1.contiguous()
        rowmax, rowmean = torch.empty(H, c, device=dev), torch.empty(H, c, device=dev)
        colmax, colmean = torch.empty(W, c, device=dev), torch.empty(W, c, device=dev)
        mean = torch.empty(c, device=dev)
        ws = torch.empty(lib.spei_plane_ws_floats(H, W, c), device=dev)
        _lib.check(lib.spei_plane_stats(tp(x1), None, 0, H, W, c, tp(rowmax), tp(rowmean), tp(colmax), tp(colmean), tp(mean), tp(ws),
                                        ctx._stream()), "spei_plane_stats")
        p = [t.detach() for t in params]
        s, g1, g2 = _gate_maps(rowmax, rowmean, colmax, colmean, mean, p[0], p[1], p[2], p[3], p[4], tuple(p[5:9]), p[9], tuple(p[10:14]))
        out = torch.empty_like(x)
        _lib.check(lib.spei_resblock_apply(tp(x.contiguous()), tp(x1), 0, tp(s), tp(g1), tp(g2), None, tp(out), c, H, W, c, ctx._stream()),
                   "spei_resblock_apply")
        fctx.save_for_backward(x1, rowmax, rowmean, colmax, colmean, mean, s, g1, g2, *params)
        fctx.meta = (H, W)
        return out

    @staticmethod
    def backward(fctx, dout):
        x1, rowmax, rowmean, colmax, colmean, mean, s, g1, g2, *params = fctx.saved_tensors
        H, W = fctx.meta
        ctx = _ctx(dout.device)
        lib = _lib.lib()
        tp = ctx._tp
        c = x1.shape[1]
        dev = dout.device
        dout = dout.contiguous()
        # gradients of the gates: sums of dOut * x1 over x, over y, over the map
        dg1, dg2, ds = torch.empty(H, c, device=dev), torch.empty(W, c, device=dev), torch.empty(c, device=dev)
        ws = torch.empty(lib.spei_plane_ws_floats(H, W, c), device=dev)
        _lib.check(lib.spei_plane_stats(tp(dout), tp(x1), 1, H, W, c, None, tp(dg1), None, tp(dg2), tp(ds), tp(ws), ctx._stream()),
                   "spei_plane_stats")
        # through the (tiny) gate maps with torch.autograd: statistics and parameters are the leaves
        with torch.enable_grad():
            stats = [t.detach().requires_grad_(True) for t in (rowmax, rowmean, colmax, colmean, mean)]
            prm = [t.detach().requires_grad_(t.is_floating_point()) for t in params]
            s2, g1b, g2b = _gate_maps(*stats, prm[0], prm[1], prm[2], prm[3], prm[4], tuple(prm[5:9]), prm[9], tuple(prm[10:14]))
            leaves = stats + [t for t in prm if t.requires_grad]
            grads = torch.autograd.grad([s2, g1b, g2b], leaves, [ds, dg1, dg2], allow_unused=True)
        d_rowmax, d_rowmean, d_colmax, d_colmean, d_mean = (g.contiguous() if g is not None else torch.zeros_like(t)
                                                            for g, t in zip(grads[:5], stats))
        dx1 = torch.empty_like(x1)
        _lib.check(lib.spei_resblock_apply_bwd(tp(dout), tp(x1), tp(s), tp(g1), tp(g2), tp(rowmax), tp(colmax), tp(d_rowmax), tp(d_rowmean),
                                               tp(d_colmax), tp(d_colmean), tp(d_mean), tp(dx1), H, W, c, ctx._stream()),
                   "spei_resblock_apply_bwd")
        pg = iter(grads[5:])
        dparams = [next(pg) if t.requires_grad else None for t in prm]
        # running statistics of the BatchNorm are buffers: no gradient
        for i in (7, 8, 12, 13):
            dparams[i] = None
        return (dout, dx1, None, None) + tuple(dparams)


def resblock_train(x: torch.Tensor, blk, H: int, W: int) -> torch.Tensor:
    """x + SE(x1) + TE(x1), x1 = conv5(relu(conv5(x)))   (model/block.py:127-140) on pixel rows; `blk`: a speinet._ResBlock."""
    t = _Conv.apply(x, blk.main[0].main[0].weight, blk.main[0].main[0].bias, H, W, 5, 1, True)
    x1 = _Conv.apply(t, blk.main[1].main[0].weight, blk.main[1].main[0].bias, H, W, 5, 1, False)
    cw, hc = blk.te.cw.conv, blk.te.hc.conv
    return _GatedSum.apply(x, x1, H, W, blk.se.fc[0].weight, blk.se.fc[0].bias, blk.se.fc[2].weight, blk.se.fc[2].bias,
                           cw.conv.weight, cw.bn.weight, cw.bn.bias, cw.bn.running_mean, cw.bn.running_var,
                           hc.conv.weight, hc.bn.weight, hc.bn.bias, hc.bn.running_mean, hc.bn.running_var)


def encoder_train(frame: torch.Tensor, recons_net):
    """Differentiable encoder_second(encoder_first(inBlock(frame))) (model/recons_video_ori.py:79-81 as used by
    model/speinet.py:82-83): frame [3,H,W] on the device -> (lv1 [H*W,32], lv2 [H/2*W/2,64], lv3 [H/4*W/4,128]) pixel rows;
    gradients flow into the parameters of `recons_net` (a speinet._Recons)."""
    if not frame.is_cuda:
        raise RuntimeError("speinet_amd.autograd runs on MI355X only (HIP kernels); there is no CPU path")
    _lib.lib()
    c, H, W = frame.shape
    with torch.cuda.device(frame.device):
        st = recons_net.inBlock
        f = _ConvIn.apply(frame, st[0][0].weight, st[0][0].bias)
        for i in range(1, len(st)):
            f = resblock_train(f, st[i], H, W)
        lv1 = f
        outs = [lv1]
        h, w = H, W
        for stage in (recons_net.encoder_first, recons_net.encoder_second):
            f = _Conv.apply(f, stage[0][0].weight, stage[0][0].bias, h, w, 5, 2, True)
            h, w = (h + 4 - 5) // 2 + 1, (w + 4 - 5) // 2 + 1
            for i in range(1, len(stage)):
                f = resblock_train(f, stage[i], h, w)
            outs.append(f)
        return tuple(outs)
