"""The per-sequence forward pass as a sequence of HIP kernel launches (one sample at a time).

Mirrors reference model/speinet.py:75-148 (`_process`, `_decode`, `_forwardbs`, `_forwardb`); every step is a
call into the C-ABI through a `speinet_amd.ops.Ctx` (the call's arithmetic mode / device; no global state).  Feature maps stay NHWC fp32 in HBM between kernels; channel
concatenations are never materialised (two-source GEMM operands, strided output views).
"""
from __future__ import annotations

import torch

from .ops import ACT_GELU, ACT_RELU, CONV_T, BMap, Ctx, FMap


# ---- encoder / decoder stacks (reference model/recons_video_ori.py:26-77) -----------------------------
def _resblocks(ctx: Ctx, f: FMap, blocks, extra=None, out=None) -> FMap:
    n = len(blocks)
    fuse = ctx.apply_fused_available(f.C)            # every block but the last hands its gated sum to the next block's first conv
    for i, pk in enumerate(blocks):
        last = i == n - 1
        f = ctx.resblock(f, pk, extra=extra if last else None, out=out if last else None, defer=fuse and not last)
    return f


def in_block(ctx: Ctx, frame: torch.Tensor, pk: dict) -> FMap:
    ctx = ctx.for_stage("enc")
    return _resblocks(ctx, ctx.conv5_in(frame, pk["head_w"], pk["head_b"]), pk["blocks"])


def enc_stage(ctx: Ctx, f: FMap, pk: dict, extra=None, out=None) -> FMap:
    ctx = ctx.for_stage("enc")
    n = pk["head_b"].numel()
    f = ctx.igemm(f, pk["head_w"], pk["head_b"], n, ksize=5, stride=2, act=ACT_RELU)
    return _resblocks(ctx, f, pk["blocks"], extra, out)


def enc(ctx: Ctx, frame: torch.Tensor, P: dict, extra=None, out=None) -> FMap:
    """encoder_second(encoder_first(inBlock(frame)))  [+ extra, fused into the last ResBlock's apply]."""
    return enc_stage(ctx, enc_stage(ctx, in_block(ctx, frame, P["inBlock"]), P["encoder_first"]), P["encoder_second"], extra, out)


def enc_batched(ctx: Ctx, frames: list, P: dict):
    """The three encoder stages on ALL of a frame's passes at once: frames = [3,H,W] tensors (the window's frames, their
    Richardson-Lucy priors, the sharp reference; model/speinet.py:82-83,125-131 run the same recons_net stacks on each).  Every
    layer is one launch over the stacked maps (ops.BMap); per map the kernels, tiles and arithmetic are those of `enc` — the
    results are bit-identical to one pass at a time.  Returns the three levels as BMaps."""
    ctx = ctx.for_stage("enc")
    assert ctx.batched_available()
    b, (h, w) = len(frames), frames[0].shape[-2:]
    ib = P["inBlock"]
    lv1 = BMap.empty(b, h, w, ib["head_b"].numel(), frames[0].device)
    lib_in = lambda i: ctx.conv5_in(frames[i], ib["head_w"], ib["head_b"], out=lv1.map(i))     # persistent, full-chip launches already
    for i in range(b):
        lib_in(i)
    for pk in ib["blocks"]:
        lv1 = ctx.resblock_batched(lv1, pk)
    levels = [lv1]
    f = lv1
    for name in ("encoder_first", "encoder_second"):
        st = P[name]
        f = ctx.igemm_batched(f, st["head_w"], st["head_b"], st["head_b"].numel(), 5, stride=2, act=ACT_RELU)
        for pk in st["blocks"]:
            f = ctx.resblock_batched(f, pk)
        levels.append(f)
    return levels


def dec_stage(ctx: Ctx, f: FMap, pk: dict) -> FMap:
    f = _resblocks(ctx, f, pk["blocks"])
    return ctx.for_stage("convt").igemm(f, pk["tail_w"], pk["tail_b"], pk["tail_b"].numel(), ksize=3, stride=2, mode=CONV_T, act=ACT_RELU)


# ---- cross-window-attention SwinIR (reference model/swinir.py:763-810) ---------------------------------
class SwinX:
    """x-side tensors shared by the two swin calls of one frame: conv_first(f_mid) and its patch-embed LN."""

    def __init__(self, ctx: Ctx, f_mid: FMap, sw: dict):
        ctx = ctx.for_stage("swin")
        self.f_mid = f_mid
        self.x_first = ctx.igemm(f_mid, sw["conv_first_w"], sw["conv_first_b"], 256, ksize=3)
        self.xt0 = ctx.layernorm(self.x_first.t, sw["pe_g"], sw["pe_b"])


def swin(ctx: Ctx, sx: SwinX, feat: FMap, sw: dict, out: FMap) -> FMap:
    ctx = ctx.for_stage("swin")
    h, w = feat.H, feat.W
    m = h * w
    dev = feat.t.device
    y_first = ctx.igemm(feat, sw["conv_first_w"], sw["conv_first_b"], 256, ksize=3)
    yt = ctx.layernorm(y_first.t, sw["pe_g"], sw["pe_b"])
    idt = ctx.inter_dtype()                       # bf16 in the throughput mode: these tensors only feed GEMMs / attention
    yhat = ctx.layernorm(yt, out_dtype=idt)       # norm1(y) without affine; gamma/beta live in wq/bq (pack.py)
    del y_first, yt
    r = sx.xt0.clone()                            # RSTB input / running residual
    bufs = [torch.empty(m, 256, device=dev), torch.empty(m, 256, device=dev)]
    xh = torch.empty(m, 256, device=dev, dtype=idt)
    q = torch.empty(m, 256, device=dev, dtype=idt)
    kv = torch.empty(m, 512, device=dev, dtype=idt)
    att = torch.empty(m, 256, device=dev, dtype=idt)
    hid = torch.empty(m, 512, device=dev, dtype=idt)
    for layer in sw["layers"]:
        cur = r
        for bi, bk in enumerate(layer["blocks"]):
            shift = 0 if bi % 2 == 0 else 2
            nxt = bufs[bi % 2]
            if ctx.attn_fused_available():
                ctx.attn_fused(cur, yhat, bk, h, w, shift, out=nxt)               # whole attention branch, one kernel
            else:
                if ctx.ln_fused_available():
                    ctx.linear(cur, bk["wkv"], bk["bkv"], out=kv, ln_input=True)  # norm1 inside the GEMM's staging
                else:
                    ctx.layernorm(cur, out=xh)
                    ctx.linear(xh, bk["wkv"], bk["bkv"], out=kv)
                ctx.linear(yhat, bk["wq"], bk["bq"], out=q)
                ctx.window_attention(q, kv, bk["relbias"], h, w, shift, out=att)
                ctx.linear(att, bk["wproj"], bk["bproj"], residual=cur, out=nxt)
            if ctx.mlp_fused_available():
                ctx.mlp_fused(nxt, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out=nxt)
            else:
                if ctx.ln_fused_available():
                    ctx.linear(nxt, bk["w1"], bk["b1"], act=ACT_GELU, out=hid, ln_input=True)
                else:
                    ctx.layernorm(nxt, out=xh)
                    ctx.linear(xh, bk["w1"], bk["b1"], act=ACT_GELU, out=hid)
                ctx.linear(hid, bk["w2"], bk["b2"], residual=nxt, out=nxt)
            cur = nxt
        # RSTB: conv3x3(blocks(x)) + x   (swinir.py:483-484), in place on the residual buffer
        rf = FMap(r, h, w, 256)
        ctx.igemm(FMap(cur, h, w, 256), layer["conv_w"], layer["conv_b"], 256, ksize=3, residual=rf, out=rf)
    xt = ctx.layernorm(r, sw["norm_g"], sw["norm_b"])
    res = ctx.igemm(FMap(xt, h, w, 256), sw["cab_w"], sw["cab_b"], 256, ksize=3, residual=sx.x_first)
    return ctx.igemm(res, sw["conv_last_w"], sw["conv_last_b"], 128, ksize=3, residual=sx.f_mid, out=out)


def swin_multi(ctx: Ctx, sx: SwinX, feats: BMap, sw: dict, outs: list) -> None:
    """Every Swin call of a frame — `self.swin(f_mid, features)` once per neighbour frame, model/speinet.py:84: the same weights, the
    same x-side input, one y per call — as ONE batch: `feats` holds the calls' y maps stacked ([B*H*W, 128]), every kernel of the 36
    blocks runs once over the stacked token maps (the attention kernel per map inside one launch, the MLP / LayerNorm row-wise, the
    3x3 convolutions with gridDim.y = map).  Round 3 ran the calls side by side on two streams: every launch then filled the chip
    with ITS OWN workgroups and the pair gained little (9.7 ms for 10.2 ms of kernels); batched, a launch has twice the workgroups
    (4.5 rounds of the four-window attention kernel instead of 2.25) and the frame half the launches.  outs[b]: where call b's
    [H*W, 128] result goes (views of the `fusion` input)."""
    ctx = ctx.for_stage("swin")
    B, h, w = feats.B, feats.H, feats.W
    m = h * w
    dev = feats.t.device
    y_first = ctx.igemm_batched(feats, sw["conv_first_w"], sw["conv_first_b"], 256, 3)
    yt = ctx.layernorm(y_first.t, sw["pe_g"], sw["pe_b"])
    idt = ctx.inter_dtype()
    yhat = ctx.layernorm(yt, out_dtype=idt)       # norm1(y) without affine; gamma/beta live in wq/bq (pack.py)
    del y_first, yt
    r = sx.xt0.repeat(B, 1)                       # RSTB input / running residual, per call
    bufs = [torch.empty(B * m, 256, device=dev), torch.empty(B * m, 256, device=dev)]
    for layer in sw["layers"]:
        cur = r
        for bi, bk in enumerate(layer["blocks"]):
            shift = 0 if bi % 2 == 0 else 2
            nxt = bufs[bi % 2]
            ctx.attn_fused(cur, yhat, bk, h, w, shift, out=nxt)
            ctx.mlp_fused(nxt, bk["w1"], bk["b1"], bk["w2"], bk["b2"], out=nxt)
            cur = nxt
        # RSTB: conv3x3(blocks(x)) + x   (swinir.py:483-484), in place on the residual buffer
        rb = BMap(r, B, h, w, 256)
        ctx.igemm_batched(BMap(cur, B, h, w, 256), layer["conv_w"], layer["conv_b"], 256, 3, residual=rb, out=rb)
    xt = ctx.layernorm(r, sw["norm_g"], sw["norm_b"])
    x_first = BMap(sx.x_first.t.repeat(B, 1), B, h, w, 256)
    res = ctx.igemm_batched(BMap(xt, B, h, w, 256), sw["cab_w"], sw["cab_b"], 256, 3, residual=x_first)
    for b in range(B):                            # strided outputs (slices of the 384-channel fusion input): one launch per call
        ctx.igemm(res.map(b), sw["conv_last_w"], sw["conv_last_b"], 128, ksize=3, residual=sx.f_mid, out=outs[b])


# ---- SearchTransfer / SelfTransfer (reference model/SearchTransfer.py) ---------------------------------
# Each is split around the correlation arg-max kernel (the dominant kernel of the path): `*_plan` prepares it, the caller
# launches `plan` (engine._tail does; bench.py places that launch between two captured graph segments to time it live with
# HIP events), `*_finish` consumes what it decided.
def search_plan(ctx: Ctx, f_fusion: FMap, lv3: FMap):
    ctx = ctx.for_stage("search")
    return ctx.corr_plan(f_fusion, lv3, ctx.patch_invnorm(f_fusion), ctx.patch_invnorm(lv3))


def search_finish(ctx: Ctx, plan, f_fusion: FMap, lv1: FMap, lv2: FMap, lv3: FMap, return_arg=False):
    ctx = ctx.for_stage("search")
    s, arg = plan.s, plan.arg
    if ctx.capture is not None:
        ctx.capture.update(arg=arg, s=s)
    h3, w3 = f_fusion.H, f_fusion.W
    t3 = ctx.gather_fold(lv3, arg, h3, w3, lv3.H, lv3.W, 1)
    t2 = ctx.gather_fold(lv2, arg, h3, w3, lv3.H, lv3.W, 2)
    t1 = ctx.gather_fold(lv1, arg, h3, w3, lv3.H, lv3.W, 4)
    if return_arg:
        return s, t3, t2, t1, arg
    return s, t3, t2, t1


def search_transfer(ctx: Ctx, f_fusion: FMap, lv1: FMap, lv2: FMap, lv3: FMap, return_arg=False):
    plan = search_plan(ctx, f_fusion, lv3)
    plan.launch()
    return search_finish(ctx, plan, f_fusion, lv1, lv2, lv3, return_arg)


def self_plan(ctx: Ctx, f_fusion: FMap):
    ctx = ctx.for_stage("search")
    ref = ctx.rot90(f_fusion)
    return ctx.corr_plan(f_fusion, ref, ctx.patch_invnorm(f_fusion), ctx.patch_invnorm(ref))


def self_finish(ctx: Ctx, plan, f_fusion: FMap, P: dict):
    ctx = ctx.for_stage("search")
    s = plan.s
    if ctx.capture is not None:
        ctx.capture.update(s_self=s)
    p1, p2 = P["SelfTransfer.search1"], P["SelfTransfer.search2"]
    t2 = ctx.up_conv1x1_relu(f_fusion, p1["w"], p1["b"], 64)
    t1 = ctx.up_conv1x1_relu(t2, p2["w"], p2["b"], 32)
    return s, f_fusion, t2, t1


def self_transfer(ctx: Ctx, f_fusion: FMap, P: dict):
    plan = self_plan(ctx, f_fusion)
    plan.launch()
    return self_finish(ctx, plan, f_fusion, P)


# ---- decode (reference model/speinet.py:92-120) -----------------------------------------------------------
def decode(ctx: Ctx, ff: FMap, s: torch.Tensor, t3: FMap, t2: FMap, t1: FMap, P: dict, out: torch.Tensor) -> torch.Tensor:
    ctx = ctx.for_stage("decode")
    g = ctx.for_stage("glue")        # the 1x1 / 3x3 convs between the stacks (conv_lv*, search*; 1 % of the frame's FLOPs) at H/4, H/2
    g1 = ctx.for_stage("glue1")      # ... and at full resolution
    c = lambda name: (P[name]["w"], P[name]["b"])
    h3, w3 = ff.H, ff.W
    smap = FMap(s.view(h3 * w3, 1), h3, w3, 1)
    f_lv3 = g.igemm(ff, *c("conv_lv3"), 128, a1=t3, rowscale=s, residual=ff)
    dec2 = dec_stage(ctx.for_stage("dec2"), f_lv3, P["decoder_second"])
    s2 = ctx.upsample(smap, 2).t.view(-1)
    f_lv2 = g.igemm(dec2, *c("conv_lv2"), 64, a1=t2, rowscale=s2, residual=dec2)
    s1 = g.up_conv1x1_relu(f_lv3, *c("search1"), 64)
    sr2 = g.igemm(f_lv2, *c("search3"), 64, ksize=3, act=ACT_RELU)
    f_v3 = g.igemm(dec2, *c("search2"), 64, a1=s1, act=ACT_RELU, residual=dec2)
    f_lv2 = g.igemm(f_lv2, *c("search2"), 64, a1=sr2, act=ACT_RELU, residual=f_lv2)
    dec1 = dec_stage(ctx.for_stage("dec1"), f_lv2, P["decoder_first"])
    s4 = ctx.upsample(smap, 4).t.view(-1)
    f_lv1 = g1.igemm(dec1, *c("conv_lv1"), 32, a1=t1, rowscale=s4, residual=dec1)
    s13 = g.up_conv1x1_relu(f_v3, *c("search13"), 32)
    s23 = g1.igemm(ctx.upsample(f_lv2, 2), *c("search33"), 32, ksize=3, act=ACT_RELU)
    s33 = g1.igemm(f_lv1, *c("search43"), 32, ksize=3, act=ACT_RELU)
    acc = g1.igemm(s13, *c("search33"), 32, ksize=3, a1=s23, act=ACT_RELU, residual=f_lv1)
    g1.igemm(s13, *c("search33"), 32, ksize=3, a1=s33, act=ACT_RELU, residual=acc, out=acc)
    g1.igemm(s23, *c("search33"), 32, ksize=3, a1=s33, act=ACT_RELU, residual=acc, out=acc)
    ob = P["outBlock"]
    ctx = ctx.for_stage("out")
    f = _resblocks(ctx, acc, ob["blocks"])
    return ctx.for_stage("tail").conv5_out(f, ob["tail_w"], ob["tail_b"], out, ob.get("tail_w32"), ob.get("tail_b32"))


# ---- one sample --------------------------------------------------------------------------------------------
# The neighbour-frame branches (2 encoder passes + one swin each) and the reference-frame encoder are independent of
# each other: with more than one lane they are issued round-robin on HIP side streams (fork after conv_first(f_mid), join
# before the `fusion` conv).  Every kernel of the path leaves CUs idle in its last partial round of workgroups and in its
# load / store phases; a second stream's kernels fill those.  Captured into the frame's hipGraph as parallel branches.
# `sides`: the side streams (owned by the calling model, per device); the current stream of ctx.device is the main lane.
def _lanes(ctx: Ctx, sides) -> list:
    return [torch.cuda.current_stream(ctx.device)] + list(sides or ())


def forward_sample(ctx: Ctx, x: torch.Tensor, P: dict, n_seq: int, has_ref: bool, out: torch.Tensor, sides=()) -> torch.Tensor:
    """x [n_seq+2, 3, H, W] (contiguous, cuda) -> out [3, H, W]."""
    return _run(forward_sample_steps(ctx, x, P, n_seq, has_ref, out, sides), out)


def _run(steps, out):
    for plan in steps:           # one yield: the prepared correlation arg-max
        plan.launch()
    return out


def forward_sample_steps(ctx: Ctx, x: torch.Tensor, P: dict, n_seq: int, has_ref: bool, out: torch.Tensor, sides=()):
    """Generator form of `forward_sample`: runs everything up to the correlation arg-max, yields its prepared launch (the caller
    launches it), then runs the rest.  All side-stream work is joined before the yield.

    f_mid, the neighbour-frame swin fusions and the 1x1 `fusion` conv (speinet.py:75-90,129-134), then
    SearchTransfer / SelfTransfer and the decoder (:92-148)."""
    h, w = x.shape[-2:]
    h3, w3 = h // 4, w // 4
    dev = x.device
    lanes = _lanes(ctx, sides)
    main = lanes[0]
    if ctx.for_stage("enc").batched_available():
        yield from forward_batch_steps(ctx, x[None], P, n_seq, [not has_ref], out[None], sides)
        return
    for s_ in lanes[1:]:
        s_.wait_stream(main)                      # fork: the input frames (and anything before them) are ready
    lv = None
    if has_ref:                                   # sharp-reference pyramid: last lane, ahead of its neighbour frames
        with torch.cuda.stream(lanes[-1]):
            lv = reference_pyramid(ctx, x[n_seq + 1], P)
    cat = FMap(torch.empty(h3 * w3, 128 * n_seq, device=dev), h3, w3, 128 * n_seq)
    mid = x[n_seq // 2]
    e0 = enc(ctx, mid, P)
    f_mid = enc(ctx, ctx.rl_prior(mid, 5, 0.01), P, extra=e0, out=cat.view(0, 128))
    others = [i for i in range(n_seq) if i != n_seq // 2]
    if ctx.for_stage("swin").swin_multi_available():
        # the neighbour frames' encoder passes on the lanes, then ALL their swin calls as one batch (swin_multi)
        fb = BMap.empty(len(others), h3, w3, 128, dev)
        for k, i in enumerate(others):
            with torch.cuda.stream(lanes[k % len(lanes)]):
                e = enc(ctx, x[i], P)
                enc(ctx, ctx.rl_prior(x[i], 1, 0.01), P, extra=e, out=fb.map(k))
                del e
        for s_ in lanes[1:]:
            main.wait_stream(s_)                  # join
        yield from _fuse_tail(ctx, f_mid, fb, cat, lv, P, out, lanes)
        return
    sx = SwinX(ctx, f_mid, P["swin"])
    ready = torch.cuda.Event()
    ready.record(main)
    for slot, i in enumerate(others, start=1):
        lane = lanes[(slot - 1) % len(lanes)]
        with torch.cuda.stream(lane):
            e = enc(ctx, x[i], P)                      # the neighbour frame's own passes need nothing from the middle frame:
            feat = enc(ctx, ctx.rl_prior(x[i], 1, 0.01), P, extra=e)
            if lane is not main:
                lane.wait_event(ready)                 # ... only its swin call does (f_mid, conv_first(f_mid))
            swin(ctx, sx, feat, P["swin"], out=cat.view(128 * slot, 128))
            del e, feat
    for s_ in lanes[1:]:
        main.wait_stream(s_)                      # join
    yield from _tail(ctx, cat, lv, P, out)


def forward_batch_steps(ctx: Ctx, x: torch.Tensor, P: dict, n_seq: int, zero_ref: list, out: torch.Tensor, sides=(), max_maps: int = 16):
    """x [B, n_seq+2, 3, H, W] -> out [B, 3, H, W] with ALL encoder passes of a group of samples (7 per sample with a sharp reference,
    6 without; up to `max_maps` maps) in one launch per layer (enc_batched; the reference batches the same way, model/speinet.py:
    150-168 runs `_forwardbs` / `_forwardb` on the sub-batches).  Yields one prepared correlation launch per sample, in sample order;
    per sample the swin calls, the search and the decoder are those of `forward_sample`.  Bit-identical to one sample at a time."""
    B = x.shape[0]
    h, w = x.shape[-2:]
    h3, w3 = h // 4, w // 4
    dev = x.device
    lanes = _lanes(ctx, sides)
    main = lanes[0]
    others = [i for i in range(n_seq) if i != n_seq // 2]
    b0 = 0
    while b0 < B:
        # samples of this group: as many as fit the map budget
        group, maps = [], 0
        while b0 + len(group) < B:
            need = 2 + 2 * len(others) + (0 if zero_ref[b0 + len(group)] else 1)
            if group and maps + need > max_maps:
                break
            group.append(b0 + len(group))
            maps += need
        frames, first = [], {}
        for b in group:
            first[b] = len(frames)
            mid = x[b, n_seq // 2]
            frames += [mid, ctx.rl_prior(mid, 5, 0.01)]
            for i in others:
                frames += [x[b, i], ctx.rl_prior(x[b, i], 1, 0.01)]
            if not zero_ref[b]:
                frames.append(x[b, n_seq + 1])
        lv1, lv2, lv3 = enc_batched(ctx, frames, P)
        del frames
        for b in group:
            m0 = first[b]
            cat = FMap(torch.empty(h3 * w3, 128 * n_seq, device=dev), h3, w3, 128 * n_seq)
            f_mid = FMap(ctx.add(lv3.map(m0 + 1).t, lv3.map(m0).t), h3, w3, 128)            # enc(RL5(mid)) + enc(mid)   (speinet.py:130-132)
            cat.t[:, :128].copy_(f_mid.t)
            fb = BMap.empty(len(others), h3, w3, 128, dev)                                   # enc(RL1(x_i)) + enc(x_i), the calls' y maps stacked
            for k in range(len(others)):
                ctx.add(lv3.map(m0 + 3 + 2 * k).t, lv3.map(m0 + 2 + 2 * k).t, out=fb.map(k).t)
            mr = m0 + 2 + 2 * len(others)
            lv = None if zero_ref[b] else (lv1.map(mr), lv2.map(mr), lv3.map(mr))
            yield from _fuse_tail(ctx, f_mid, fb, cat, lv, P, out[b], lanes)
        b0 += len(group)


def _fuse_tail(ctx: Ctx, f_mid: FMap, fb: BMap, cat: FMap, lv, P: dict, out: torch.Tensor, lanes: list):
    """The neighbour-frame fusions into `cat` (its first 128 channels already hold f_mid), then `_tail`.  One batched pass over the
    stacked y maps where the mode has the batched kernels (swin_multi), else one swin call per neighbour frame on the side lanes."""
    main = lanes[0]
    sx = SwinX(ctx, f_mid, P["swin"])
    if ctx.for_stage("swin").swin_multi_available():
        swin_multi(ctx, sx, fb, P["swin"], [cat.view(128 * slot, 128) for slot in range(1, fb.B + 1)])
    else:
        ready = torch.cuda.Event()
        ready.record(main)
        for slot in range(1, fb.B + 1):
            lane = lanes[(slot - 1) % len(lanes)]
            if lane is not main:
                lane.wait_event(ready)
            with torch.cuda.stream(lane):
                swin(ctx, sx, fb.map(slot - 1), P["swin"], out=cat.view(128 * slot, 128))
        for s_ in lanes[1:]:
            main.wait_stream(s_)                  # join
    yield from _tail(ctx, cat, lv, P, out)


def _tail(ctx: Ctx, cat: FMap, lv, P: dict, out: torch.Tensor):
    fw = P["fusion"]
    ff = ctx.igemm(cat, fw["w"], fw["b"], 128)
    plan = search_plan(ctx, ff, lv[2]) if lv is not None else self_plan(ctx, ff)
    yield plan
    if lv is not None:
        s, t3, t2, t1 = search_finish(ctx, plan, ff, *lv)
    else:
        s, t3, t2, t1 = self_finish(ctx, plan, ff, P)
    decode(ctx, ff, s, t3, t2, t1, P, out)


# ---- cross-window reuse --------------------------------------------------------------------
# A clip is deblurred with stride-1 sliding windows, so most encoder passes of a window repeat work of the previous
# ones: frame t is the right neighbour of window t-1, the middle of window t and the left neighbour of window t+1, and a
# sharp reference serves several windows.  The per-frame pieces below are exactly the sub-graphs `forward_sample` runs
# (same kernels, same operands => bit-identical results); `fuse_and_decode` is everything after them.
def encode_raw(ctx: Ctx, frame: torch.Tensor, P: dict) -> FMap:
    """enc(frame): shared by the frame's RL-1 (neighbour) and RL-5 (middle) sums."""
    return enc(ctx, frame, P)


def encode_sum(ctx: Ctx, frame: torch.Tensor, iters: int, e_raw: FMap, P: dict) -> FMap:
    """enc(RL_iters(frame)) + enc(frame)  (speinet.py:82-84 with iters = 5 for the middle frame, :129-132 with 1)."""
    return enc(ctx, ctx.rl_prior(frame, iters, 0.01), P, extra=e_raw)


def reference_pyramid(ctx: Ctx, frame: torch.Tensor, P: dict):
    lv1 = in_block(ctx, frame, P["inBlock"])
    lv2 = enc_stage(ctx, lv1, P["encoder_first"])
    return lv1, lv2, enc_stage(ctx, lv2, P["encoder_second"])


def fuse_and_decode(ctx: Ctx, f_mid: FMap, feats: list, lv, P: dict, n_seq: int, out: torch.Tensor, sides=()) -> torch.Tensor:
    return _run(fuse_and_decode_steps(ctx, f_mid, feats, lv, P, n_seq, out, sides), out)


def fuse_and_decode_steps(ctx: Ctx, f_mid: FMap, feats: list, lv, P: dict, n_seq: int, out: torch.Tensor, sides=()):
    """Everything after the encoders: the neighbour-frame fusions (on the side lanes), `fusion`, SearchTransfer (lv = the
    reference pyramid) or SelfTransfer (lv = None), decoder.  f_mid / feats: [H/4*W/4, 128] maps."""
    h3, w3 = f_mid.H, f_mid.W
    dev = f_mid.t.device
    lanes = _lanes(ctx, sides)
    main = lanes[0]
    cat = FMap(torch.empty(h3 * w3, 128 * n_seq, device=dev), h3, w3, 128 * n_seq)
    cat.t[:, :128].copy_(f_mid.t)
    fb = BMap(torch.cat([f.t for f in feats]), len(feats), h3, w3, 128)
    yield from _fuse_tail(ctx, f_mid, fb, cat, lv, P, out, lanes)


# ---- the `swint` variant (reference model/swint.py:51-67) ---------------------------------------------------------------------
def forward_swint(ctx: Ctx, x: torch.Tensor, P: dict, n_seq: int, out: torch.Tensor, sides=()) -> torch.Tensor:
    """x [>= n_seq, 3, H, W] -> out [3, H, W]: encoders without the RL prior, one swin call per neighbour frame, the 1x1
    `conv` over the concatenation, then the plain decoder (no SearchTransfer).  n_seq == 1: f_mid + swin(f_mid, f_mid)."""
    h, w = x.shape[-2:]
    h3, w3 = h // 4, w // 4
    dev = x.device
    lanes = _lanes(ctx, sides)
    main = lanes[0]
    for s_ in lanes[1:]:
        s_.wait_stream(main)
    cat = FMap(torch.empty(h3 * w3, 128 * n_seq, device=dev), h3, w3, 128 * n_seq)
    f_mid = enc(ctx, x[n_seq // 2], P, out=cat.view(0, 128))
    sx = SwinX(ctx, f_mid, P["swin"])
    if n_seq == 1:
        f_trans = FMap.empty(h3, w3, 128, dev)
        swin(ctx, sx, f_mid, P["swin"], out=f_trans)
        cat = FMap(ctx.add(f_mid.t, f_trans.t), h3, w3, 128)
    else:
        ready = torch.cuda.Event()
        ready.record(main)
        for slot, i in enumerate([i for i in range(n_seq) if i != n_seq // 2], start=1):
            lane = lanes[(slot - 1) % len(lanes)]
            if lane is not main:
                lane.wait_event(ready)
            with torch.cuda.stream(lane):
                swin(ctx, sx, enc(ctx, x[i], P), P["swin"], out=cat.view(128 * slot, 128))
        for s_ in lanes[1:]:
            main.wait_stream(s_)
    fw = P["conv"]
    ff = ctx.igemm(cat, fw["w"], fw["b"], 128)
    dctx = ctx.for_stage("decode")
    dec1 = dec_stage(dctx, dec_stage(dctx, ff, P["decoder_second"]), P["decoder_first"])
    ob = P["outBlock"]
    octx = dctx.for_stage("out")
    f = _resblocks(octx, dec1, ob["blocks"])
    return octx.for_stage("tail").conv5_out(f, ob["tail_w"], ob["tail_b"], out, ob.get("tail_w32"), ob.get("tail_b32"))
