"""Training step, first slice: HIP forward + backward of the recons_net encoder stack through
speinet_amd.train (the one set of autograd Functions the product ships), against (a) torch.autograd of the same op in fp64 on the CPU, per op, and (b) the REFERENCE's own
gradients (tests/golden/make_golden_grad.py, G19: its modules, its autograd) for every parameter of inBlock /
encoder_first / encoder_second.  Tolerances are fp32 round-off of re-ordered sums relative to the gradient's norm."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from speinet_amd import train as T                       # noqa: E402
from speinet_amd.speinet import SPEINet, default_args    # noqa: E402
from speinet_amd.synth import synth_frames               # noqa: E402

DEV = "cuda:0"


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32))


def rows(x):          # [1,C,H,W] -> [H*W, C]
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1]).contiguous()


def rel(a, b):
    return (a.detach().cpu().double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-30)


@pytest.mark.parametrize("cin,cout,k,stride,h,w,relu", [(32, 32, 5, 1, 20, 24, True), (64, 64, 5, 1, 13, 17, False), (32, 64, 5, 2, 40, 60, True),
                                                       (64, 128, 5, 2, 22, 18, True), (128, 128, 5, 1, 10, 15, False), (64, 32, 3, 1, 21, 19, False),
                                                       (128, 64, 1, 1, 9, 11, True)])
def test_conv_forward_backward(cin, cout, k, stride, h, w, relu):
    x, wt, b = rnd(1, 1, cin, h, w), rnd(2, cout, cin, k, k, scale=1.0 / np.sqrt(cin * k * k)), rnd(3, cout, scale=0.1)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, wt, b))
    y = F.conv2d(xd, wd, bd, stride=stride, padding=k // 2)
    y = F.relu(y) if relu else y
    g = rnd(4, *y.shape)
    y.backward(g.double())
    xg = rows(x).to(DEV).requires_grad_(True)
    wg, bg = wt.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = T._Conv2d.apply(xg, wg, bg, None, 1, h, w, k, stride, relu)
    assert rel(out, rows(y.detach().float())) < 1e-5
    out.backward(rows(g).to(DEV))
    assert rel(xg.grad, rows(xd.grad.float())) < 2e-5, "data gradient"
    assert rel(wg.grad, wd.grad.float()) < 2e-5, "weight gradient"
    assert rel(bg.grad, bd.grad.float()) < 2e-5, "bias gradient"


def test_conv_in_backward():
    x, wt, b = torch.rand(3, 28, 36), rnd(5, 32, 3, 5, 5, scale=0.1), rnd(6, 32, scale=0.1)
    wd, bd = wt.double().requires_grad_(True), b.double().requires_grad_(True)
    y = F.relu(F.conv2d(x[None].double(), wd, bd, padding=2))
    g = rnd(7, *y.shape)
    y.backward(g.double())
    wg, bg = wt.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = T._ConvIn.apply(x.to(DEV)[None], wg, bg)
    assert rel(out, rows(y.detach().float())) < 1e-5
    out.backward(rows(g).to(DEV))
    assert rel(wg.grad, wd.grad.float()) < 2e-5 and rel(bg.grad, bd.grad.float()) < 2e-5


@pytest.mark.parametrize("c,h,w", [(32, 20, 24), (64, 33, 17), (128, 10, 45)])
def test_resblock_backward_vs_torch(synth_sd, c, h, w):
    """One ResBlock (conv-relu-conv, SE + triplet gates, skip) forward and backward against a plain-torch fp64 statement of
    model/block.py:127-140 with the same parameters: output, data gradient and every parameter gradient."""
    net = SPEINet(args=default_args())
    net.load_state_dict(synth_sd, strict=True)
    stage = {32: net.recons_net.inBlock, 64: net.recons_net.encoder_first, 128: net.recons_net.encoder_second}[c]
    blk = stage[2].to(DEV)
    x = rnd(10 + c, 1, c, h, w)
    # torch fp64 statement
    ref = SPEINet(args=default_args())
    ref.load_state_dict(synth_sd, strict=True)
    rb = {32: ref.recons_net.inBlock, 64: ref.recons_net.encoder_first, 128: ref.recons_net.encoder_second}[c][2].double().eval()
    xd = x.double().requires_grad_(True)
    x1 = rb.main[1].main[0](F.relu(rb.main[0].main[0](xd)))
    y = rb.se.fc(x1.mean(dim=(2, 3))).view(1, c, 1, 1)
    xp = x1.permute(0, 3, 2, 1)
    z = torch.cat((xp.max(dim=1)[0].unsqueeze(1), xp.mean(dim=1).unsqueeze(1)), dim=1)
    g1 = rb.te.cw.conv.bn(rb.te.cw.conv.conv(z)).permute(0, 3, 2, 1)
    xp = x1.permute(0, 2, 1, 3)
    z = torch.cat((xp.max(dim=1)[0].unsqueeze(1), xp.mean(dim=1).unsqueeze(1)), dim=1)
    g2 = rb.te.hc.conv.bn(rb.te.hc.conv.conv(z)).permute(0, 2, 1, 3)
    out_ref = xd + x1 * (y + g1 + g2)
    g = rnd(20 + c, *out_ref.shape)
    out_ref.backward(g.double())
    # HIP
    xg = rows(x).to(DEV).requires_grad_(True)
    out = T.resblock(xg, blk, 1, h, w, False)       # eval-mode gates: BatchNorm(1) on its running statistics
    assert rel(out, rows(out_ref.detach().float())) < 1e-5
    out.backward(rows(g).to(DEV))
    assert rel(xg.grad, rows(xd.grad.float())) < 5e-5, "data gradient"
    refp = dict(rb.named_parameters())
    for k, p in blk.named_parameters():
        assert p.grad is not None, k
        assert rel(p.grad, refp[k].grad.float()) < 1e-4, k


@pytest.mark.parametrize("name,h,w", [("g19_enc_grad_40x60", 40, 60), ("g19_enc_grad_100x100", 100, 100)])
def test_encoder_gradients_vs_reference(golden_dir, synth_sd, name, h, w):
    """Every parameter gradient of the three encoder stages against the reference's own autograd (G19): L2 norm and a strided
    sample of each gradient; bitwise reproducible from run to run (fixed-order reductions)."""
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    seed = int(d["seed"])
    net = SPEINet(args=default_args())
    net.load_state_dict(synth_sd, strict=True)
    net = net.to(DEV)
    x = synth_frames(1, h, w, seed=seed)[0, 1].to(DEV)

    def run():
        net.zero_grad()
        lv1, lv2, lv3 = T.encoder(x[None], net.recons_net, False, pyramid=True)
        g = torch.Generator().manual_seed(seed + 1000)
        r1, r2, r3 = (torch.randn(*s, generator=g) for s in ((1, 32, h, w), (1, 64, h // 2, w // 2), (1, 128, h // 4, w // 4)))
        loss = (lv3 * rows(r3).to(DEV)).sum() + 0.5 * (lv2 * rows(r2).to(DEV)).sum() + 0.25 * (lv1 * rows(r1).to(DEV)).sum()
        loss.backward()
        return loss.item(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    loss, grads = run()
    assert abs(loss - float(d["loss"])) < 1e-3 * max(1.0, abs(float(d["loss"])))
    keys = [k[5:] for k in d.files if k.startswith("norm/")]
    assert len(keys) == 132 and set(keys) == set(grads), set(keys) ^ set(grads)
    worst, errs = 0.0, []
    for k in keys:
        g = grads[k].cpu().reshape(-1)
        n_ref = float(d["norm/" + k])
        sub = torch.from_numpy(d["sub/" + k])
        e_norm = abs(g.norm().item() - n_ref) / max(n_ref, 1e-12)
        # the sample's error is measured against the gradient's typical magnitude (norm / sqrt(numel) per element), not against the
        # sampled elements' own size: a single sampled element can be a near-cancelling sum
        e_sub = (g[::61] - sub).norm().item() / max(sub.norm().item(), n_ref * (sub.numel() / g.numel()) ** 0.5, 1e-12)
        if g.numel() > 4:
            worst = max(worst, e_norm, e_sub)
            errs.append((max(e_norm, e_sub), k))
        if g.numel() <= 4:
            # a one-number gradient (the BatchNorm(1) affine of a gate) is a sum of ~10^4 signed terms that can cancel to 1e-3 of
            # their size (values from 0.04 to 44 across blocks): its error is measured against the gradient norm of the SAME gate's
            # convolution weight (sums of the same terms), not against its own, possibly near-cancelled, value
            scale = max(n_ref, float(d["norm/" + k.rsplit(".bn.", 1)[0] + ".conv.weight"]))
            assert abs(g.norm().item() - n_ref) < 2e-3 * scale, (k, g.norm().item(), n_ref, scale)
            continue
        # 100x100: every gradient within 4e-6.  40x60: a uniform ~4e-4 on ALL 132 gradients at once — the signature of ONE ReLU /
        # arg-max decision on an element within fp32 round-off of its threshold resolving differently from the CPU's summation
        # order (one element of 76 800 moves every downstream sum by ~1/sqrt(numel)), not of an arithmetic difference
        assert e_norm < 2e-3 and e_sub < 2e-3, (k, e_norm, e_sub)
    print(f"{name}: 132 parameter gradients, worst relative deviation from the reference {worst:.1e}; largest: "
          + ", ".join(f"{k[11:]} {e:.1e}" for e, k in sorted(errs, reverse=True)[:6]))
    loss2, grads2 = run()
    assert loss2 == loss and all(torch.equal(grads[k], grads2[k]) for k in grads), "gradients are not bitwise reproducible"
