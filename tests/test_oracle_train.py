"""The oracle's train-mode graph (`oracle.speinet_oracle.train_mode`: batch-statistics BatchNorm(1) + DropPath factors; gradients
from torch autograd over the functional restatement) pinned against the reference's own training step: G20 (model/swint.py, 40x40,
B = 2).  With this pin the restatement serves as the checker of the HIP training step at sizes no fixture holds
(tests/test_gpu_train.py) and as the CPU baseline of tools/train_bench.py.  The full model (search, both routing branches, cross-scale decoder) is pinned the same way through G21."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import speinet_oracle as O                                   # noqa: E402
from speinet_amd.loss import Loss                                        # noqa: E402
from speinet_amd.synth import state_dict_template, synth_frames, synth_state_dict   # noqa: E402


def _scales(draws, depths, n_calls):
    from speinet_amd.train import drop_path_rates
    it = iter(torch.from_numpy(draws))
    return [[None if r <= 0 else (next(it), next(it)) for r in drop_path_rates(depths)] for _ in range(n_calls)]


def oracle_train_step(kind, x, gt, sd, scales, seed, cfg):
    """loss and {name: gradient} of one training forward / backward of the restatement (float tensors of sd are the leaves)."""
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    np.random.seed(seed)
    with O.train_mode(scales):
        out = (O.forward_swint if kind == "swint" else O.forward)(x, leaves, cfg)
    loss = Loss("1*L1+2*HEM", device="cpu")(out, gt)
    loss.backward()
    return out.detach(), loss.item(), {k: v.grad for k, v in leaves.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


def test_oracle_train_mode_vs_reference_training_step(golden_dir):
    """In FLOAT64 the functional restatement and the reference module are the same arithmetic up to summation order: output, loss and
    every gradient agree with the fixture's float64 run to 1e-8 of the gradient norm.  (In fp32 the two CPU evaluations differ by up
    to 1.6e-2 on the cancellation-prone gate weights — BatchNorm through a library kernel vs tensor arithmetic — which is the noise
    floor any fp32 comparison of this graph lives on; the HIP path sits at 2e-3.)"""
    torch.set_num_threads(8)
    d = np.load(os.path.join(golden_dir, "g20_train_swint_40x40.npz"))
    seed, n_seq, b, h, w = (int(d[k]) for k in ("seed", "n_sequence", "b", "h", "w"))
    from speinet_amd.swint import SPEINet
    from speinet_amd.speinet import default_args
    args = default_args()
    args.n_sequence = n_seq
    net = SPEINet(n_sequence=n_seq, args=args)
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in synth_state_dict(net.state_dict(), seed=0).items()}
    cfg = O.Cfg(n_sequence=n_seq)
    x = synth_frames(b, h, w, seed=seed)[:, :n_seq].contiguous().double()
    gt = synth_frames(b, h, w, seed=seed + 500)[:, 1].contiguous().double()
    out, loss, grads = oracle_train_step("swint", x, gt, sd, _scales(d["draws"], cfg.depths, n_seq - 1), seed, cfg)
    assert (out.float() - torch.from_numpy(d["out"])).abs().max().item() < 2e-5           # the fixture's output is the fp32 run's
    assert abs(loss - float(d["loss64"])) < 1e-9
    gmax = max(float(d[k]) for k in d.files if k.startswith("norm64/"))
    worst = (0.0, "")
    for k in (f[7:] for f in d.files if f.startswith("norm64/")):
        g = grads[k].reshape(-1)
        scale = max(float(d["norm64/" + k]), 1e-5 * gmax)
        e = max((g[::97].float() - torch.from_numpy(d["sub64/" + k])).norm().item(), abs(g.norm().item() - float(d["norm64/" + k]))) / scale
        worst = max(worst, (e, k))
    print(f"oracle (float64) training gradients vs the reference's float64 run: worst deviation {worst[0]:.1e} of the gradient norm ({worst[1]})")
    assert worst[0] < 1e-6          # sub64 is stored as float32: 6e-8 relative per element


def test_oracle_train_mode_full_model_vs_reference(golden_dir):
    """The same for `model/speinet.py` (G21: three 40x40 windows, the second without a reference): SearchTransfer's max / gather and
    SelfTransfer under autograd, the cross-scale decoder, BatchNorm statistics per routing class — float64 against float64."""
    torch.set_num_threads(8)
    d = np.load(os.path.join(golden_dir, "g21_train_speinet_40x40.npz"))
    seed, b, h, w = (int(d[k]) for k in ("seed", "b", "h", "w"))
    from speinet_amd.speinet import SPEINet, default_args
    from speinet_amd.train import drop_path_rates
    net = SPEINet(args=default_args())
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in synth_state_dict(net.state_dict(), seed=0).items()}
    cfg = O.Cfg(n_sequence=3)
    x = synth_frames(b, h, w, seed=seed, zero_ref=(1,)).contiguous().double()
    gt = synth_frames(b, h, w, seed=seed + 500)[:, 1].contiguous().double()
    it = iter(zip(d["draws"], d["draw_len"]))
    calls = []
    for _ in range(4):                                   # 2 swin calls of the no-reference sub-batch (1 sample), then 2 of the other (2)
        call = []
        for r in drop_path_rates(cfg.depths):
            if r <= 0:
                call.append(None)
            else:
                (r0, n0), (r1, n1) = next(it), next(it)
                call.append((torch.from_numpy(r0[:int(n0)]), torch.from_numpy(r1[:int(n1)])))
        calls.append(call)
    out, loss, grads = oracle_train_step("speinet", x, gt, sd, calls, seed, cfg)
    assert (out.float() - torch.from_numpy(d["out"])).abs().max().item() < 2e-5
    # (the fixture's float64 run received the Richardson-Lucy prior frames in fp32 — model/rcl.py builds fp32 kernels — this one
    # computes them in float64: inputs differ by 1e-8, the loss by 3e-9)
    assert abs(loss - float(d["loss64"])) < 1e-7
    gmax = max(float(d[k]) for k in d.files if k.startswith("norm64/"))
    worst = (0.0, "")
    for k in (f[7:] for f in d.files if f.startswith("norm64/")):
        g = grads[k].reshape(-1)
        scale = max(float(d["norm64/" + k]), 1e-5 * gmax)
        e = max((g[::97].float() - torch.from_numpy(d["sub64/" + k])).norm().item(), abs(g.norm().item() - float(d["norm64/" + k]))) / scale
        worst = max(worst, (e, k))
    params = {k for k, _ in net.named_parameters()}
    assert set(str(u) for u in d["unused"]) == {k for k in params if k not in grads}          # search23, connect, SearchTransfer's convs
    print(f"oracle (float64) full-model training gradients vs the reference's float64 run: worst deviation {worst[0]:.1e} ({worst[1]})")
    assert worst[0] < 1e-4          # (the 1e-8 prior difference above, amplified on the cancelling BatchNorm scalars: 1.4e-5 measured)
