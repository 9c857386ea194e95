#!/usr/bin/env python3
"""G20 / G21: one TRAINING step of the REFERENCE's `model/swint.py` (G20) and of `model/speinet.py` itself (G21, a batch that takes
both routing branches) on the CPU, the way trainer/trainer_swint.py:34-44 runs it:
train() mode (BatchNorm2d(1) batch statistics in the ResBlock gates, DropPath in the Swin blocks), loss 1*L1 + 2*HEM
(option/template.py:11, Loss/hard_example_mining.py loaded from the reference by path), loss.backward(), Adam(lr 1e-4).step().
Synthetic name-keyed weights (seed 0), seeded inputs and targets.

`timm.models.layers.DropPath` is a dependency the reference does not pin and this image does not have: the class below restates
its published algorithm (per call: new_empty((B,1,1)).bernoulli_(keep) / keep) and RECORDS every draw, so the fixture carries
the factors the reference actually used; speinet_amd.train.drop_path_scales draws the same stream from the same seed.

Committed per case: the full output and the loss terms, the DropPath draws, per-parameter gradient L2 norms and every 97th
element of each gradient, the BatchNorm running buffers after the step, and every 97th element of every 25th parameter after
the Adam step.  The same step is then repeated in FLOAT64 (same weights, inputs, DropPath factors, HEM shuffle): `sub64/*`
holds the same gradient elements from that run, so a test can tell fp32 summation noise (how far the reference's own fp32
gradients sit from the fp64 ones) from an arithmetic difference.

Run:  python tests/golden/make_golden_train.py      (needs /root/reference; writes tests/golden/g20_train_*.npz)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, import_reference, template_args      # noqa: E402

STRIDE = 97
DRAWS = []
REPLAY = []          # when non-empty: the float64 re-run takes its factors from here instead of drawing


class DropPath(torch.nn.Module):
    """timm.models.layers.DropPath (drop_path with scale_by_keep=True), recording its draws."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        if REPLAY:
            return x * REPLAY.pop(0).to(x.dtype).view(shape)
        rt = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            rt.div_(keep)
        DRAWS.append(rt.view(-1).clone())
        return x * rt


def run_case(name, build, x, gt, hem_mod, seed):
    """One training step of `build()` in fp32 (recorded DropPath draws) and again in float64 (replayed draws); writes the fixture."""
    torch.manual_seed(0)
    net = build()
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=0.0)
    hem = hem_mod.HEM(device="cpu")
    DRAWS.clear()
    torch.manual_seed(seed)
    np.random.seed(seed)
    out = net(x)
    opt.zero_grad()
    l1 = torch.nn.L1Loss()(out, gt)
    lh = hem(out, gt)
    loss = 1.0 * l1 + 2.0 * lh
    loss.backward()
    res = {"seed": seed, "b": x.shape[0], "h": x.shape[-2], "w": x.shape[-1], "out": out.detach().numpy(), "loss": loss.item(),
           "l1": l1.item(), "hem": lh.item(), "draws": np.stack([np.pad(d.numpy(), (0, x.shape[0] - d.numel())) for d in DRAWS]),
           "draw_len": np.asarray([d.numel() for d in DRAWS])}
    n, unused = 0, []
    for k, p in net.named_parameters():
        if p.grad is None:
            unused.append(k)
            continue
        g = p.grad.reshape(-1)
        res["norm/" + k] = g.norm().item()
        res["sub/" + k] = g[::STRIDE].clone().numpy()
        n += 1
    res["unused"] = np.asarray(unused)
    opt.step()
    for i, (k, p) in enumerate(net.named_parameters()):
        if i % 25 == 0:
            res["adam/" + k] = p.detach().reshape(-1)[::STRIDE].clone().numpy()
    for k, v in net.state_dict().items():
        if "running_" in k or "num_batches_tracked" in k:
            res["bn/" + k] = v.clone().numpy()
    # ---- the same step in float64 ----
    net64 = build()
    net64.double().train()
    REPLAY.extend(DRAWS)
    np.random.seed(seed)
    import model.speinet as ms_
    orig_rl = ms_.r_l_per_channel
    # model/rcl.py builds its kernels in fp32: the (parameter-free) edge prior is evaluated in fp32 and cast, so the float64 run
    # sees exactly the prior frames the fp32 run saw; SPEINet.forward allocates its output in the default dtype
    ms_.r_l_per_channel = lambda img, k, it, lam: orig_rl(img.float(), k.float(), it, lam).double()
    torch.set_default_dtype(torch.float64)
    try:
        out64 = net64(x.double())
    finally:
        torch.set_default_dtype(torch.float32)
        ms_.r_l_per_channel = orig_rl
    assert not REPLAY
    loss64 = torch.nn.L1Loss()(out64, gt.double()) + 2.0 * hem_mod.HEM(device="cpu")(out64, gt.double())
    loss64.backward()
    res["loss64"] = loss64.item()
    worst = 0.0
    for k, p in net64.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        res["norm64/" + k] = g.norm().item()
        res["sub64/" + k] = g[::STRIDE].clone().numpy().astype(np.float32)
        worst = max(worst, float(np.linalg.norm(res["sub/" + k] - res["sub64/" + k]) / max(res["norm64/" + k], 1e-12)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
    print(f"{name}: loss {loss.item():.6f} (L1 {l1.item():.6f}, HEM {lh.item():.6f}), {n} parameter gradients ({len(unused)} parameters unused), "
          f"{len(DRAWS)} DropPath draws, {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.0f} KB; float64 re-run: loss {loss64.item():.8f}, "
          f"fp32 gradients' worst subsample distance from it {worst:.1e}")


def main():
    from speinet_amd.synth import synth_frames, synth_state_dict
    ms, *_ = import_reference()
    import model.swinir as sw
    import model.swint as mw
    sw.DropPath = DropPath
    spec = importlib.util.spec_from_file_location("ref_hem", os.path.join(REF, "Loss", "hard_example_mining.py"))
    hem_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hem_mod)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["swint", "speinet", "curve"]
    if "swint" in which:
        for name, seed, n_seq, b, h, w in (("g20_train_swint_40x40", 201, 3, 2, 40, 40), ("g20_train_swint_n1_40x60", 202, 1, 1, 40, 60)):
            args = template_args()
            args.n_sequence = n_seq

            def build():
                net = mw.SPEINet(in_channels=3, n_sequence=n_seq, out_channels=3, n_resblock=3, n_feat=32, device="cpu", args=args)
                net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
                return net
            x = synth_frames(b, h, w, seed=seed)[:, :n_seq].contiguous()
            gt = synth_frames(b, h, w, seed=seed + 500)[:, 1].contiguous()
            run_case(name, build, x, gt, hem_mod, seed)
    if "curve" in which:
        # G22: SURVEY.md §8(d) config 5 — the loss curve of N optimizer steps with DropPath disabled (identity), BatchNorm in train
        # mode: swint model, one fixed batch of two 40x40 windows, 1*L1 + 2*HEM, Adam(1e-4), 6 steps
        args = template_args()
        args.n_sequence = 3
        seed, b, h, w, steps = 221, 2, 40, 40, 6
        torch.manual_seed(0)
        net = mw.SPEINet(in_channels=3, n_sequence=3, out_channels=3, n_resblock=3, n_feat=32, device="cpu", args=args)
        net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
        for m in net.modules():
            if isinstance(m, DropPath):
                m.drop_prob = 0.0
        net.train()
        x = synth_frames(b, h, w, seed=seed)[:, :3].contiguous()
        gt = synth_frames(b, h, w, seed=seed + 500)[:, 1].contiguous()
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=0.0)
        hem = hem_mod.HEM(device="cpu")
        np.random.seed(seed)
        losses = []
        for _ in range(steps):
            out = net(x)
            opt.zero_grad()
            loss = torch.nn.L1Loss()(out, gt) + 2.0 * hem(out, gt)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        # the same six steps by the reference in FLOAT64 (same initial weights, same HEM draws): how far fp32 round-off alone moves
        # the reference's own curve — the scale a fp32 implementation with another summation order is judged against
        torch.manual_seed(0)
        net64 = mw.SPEINet(in_channels=3, n_sequence=3, out_channels=3, n_resblock=3, n_feat=32, device="cpu", args=args)
        net64.load_state_dict(synth_state_dict(net64.state_dict(), seed=0), strict=True)
        for m in net64.modules():
            if isinstance(m, DropPath):
                m.drop_prob = 0.0
        net64.double().train()
        opt64 = torch.optim.Adam(net64.parameters(), lr=1e-4, weight_decay=0.0)
        np.random.seed(seed)
        losses64 = []
        torch.set_default_dtype(torch.float64)
        try:
            for _ in range(steps):
                out = net64(x.double())
                opt64.zero_grad()
                loss = torch.nn.L1Loss()(out, gt.double()) + 2.0 * hem(out, gt.double())
                loss.backward()
                opt64.step()
                losses64.append(loss.item())
        finally:
            torch.set_default_dtype(torch.float32)
        np.savez_compressed(os.path.join(HERE, "g22_losscurve_swint_40x40.npz"), seed=seed, b=b, h=h, w=w, losses=np.asarray(losses),
                            losses64=np.asarray(losses64))
        print("g22_losscurve_swint_40x40: " + ", ".join(f"{v:.6f}" for v in losses))
        print("        the same in float64: " + ", ".join(f"{v:.6f}" for v in losses64))
        print("        |fp32 - float64|:    " + ", ".join(f"{abs(a - c):.2e}" for a, c in zip(losses, losses64)))
    if "speinet" in which:
        # G21: model/speinet.py itself (trainer/trainer_swint_hsa_nsf.py): three samples, the second with an all-zero frame 3 ->
        # `_forwardb` (SelfTransfer) on a sub-batch of one, `_forwardbs` (SearchTransfer) on the other two
        for name, seed, b, h, w, zero in (("g21_train_speinet_40x40", 211, 3, 40, 40, (1,)),):
            args = template_args()

            def build():
                net = ms.SPEINet(in_channels=3, n_sequence=3, out_channels=3, n_resblock=3, n_feat=32, device="cpu", args=args)
                net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
                return net
            x = synth_frames(b, h, w, seed=seed, zero_ref=zero).contiguous()
            gt = synth_frames(b, h, w, seed=seed + 500)[:, 1].contiguous()
            run_case(name, build, x, gt, hem_mod, seed)


if __name__ == "__main__":
    main()
