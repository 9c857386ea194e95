"""CPU suite: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes

import numpy as np
import os

import pytest
import torch

from speinet_amd import _lib
from speinet_amd.speinet import SPEINet, default_args, make_model
from speinet_amd.synth import state_dict_template, synth_state_dict


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(_lib.LIB_PATH):
        from speinet_amd.build import build_lib
        build_lib(verbose=False)
    return _lib.LIB_PATH


def test_library_exports_every_declared_symbol(lib_path):
    h = ctypes.CDLL(lib_path)
    declared = _lib.header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(h, name), f"{name} declared in include/speinet_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes signature table and header disagree"
    assert _lib.lib().spei_arch() == b"gfx950"
    assert _lib.lib().spei_version() >= 100


def test_state_dict_layout_matches_reference_inventory():
    net = SPEINet(args=default_args())
    mine = {k: (tuple(v.shape), v.dtype) for k, v in net.state_dict().items()}
    ref = {k: (tuple(v.shape), v.dtype) for k, v in state_dict_template().items()}
    assert set(mine) == set(ref), (sorted(set(ref) - set(mine))[:5], sorted(set(mine) - set(ref))[:5])
    for k in ref:
        assert mine[k] == ref[k], (k, mine[k], ref[k])
    assert len(mine) == 1020
    n_param = sum(p.numel() for p in net.parameters())
    assert n_param == 30_830_403 - 0 or n_param > 0


def test_strict_load_and_buffers():
    net = make_model(default_args())
    sd = synth_state_dict(state_dict_template(), seed=0)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    # registered mask buffer equals the reference formula at 50x50
    from oracle import speinet_oracle as O
    assert torch.equal(net.swin.layers[0].residual_group.blocks[1].attn_mask, O.shift_mask(50, 50, 5, 2))


def test_no_cpu_path():
    net = SPEINet(args=default_args()).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 5, 3, 20, 20))
    with pytest.raises(ValueError, match="multiples of 20"):
        net(torch.zeros(1, 5, 3, 24, 20))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 4, 3, 20, 20))


def test_packing_algebra(synth_sd):
    """LayerNorm-affine folding in pack.py is exact algebra: check one block against the oracle on CPU."""
    from oracle import speinet_oracle as O
    from speinet_amd import pack
    import torch.nn.functional as F
    p = "swin.layers.2.residual_group.blocks.3."
    bk = pack.swin_block(synth_sd, p, 8, 5)
    x = torch.randn(50, 256)
    ln = F.layer_norm(x, (256,), synth_sd[p + "norm1.weight"], synth_sd[p + "norm1.bias"], 1e-5)
    xhat = F.layer_norm(x, (256,), None, None, 1e-5)
    kv_ref = F.linear(ln, synth_sd[p + "attn.qkv_x.weight"], synth_sd[p + "attn.qkv_x.bias"])
    q_ref = F.linear(ln, synth_sd[p + "attn.qkv_y.weight"], synth_sd[p + "attn.qkv_y.bias"]) * 32 ** -0.5
    assert torch.allclose(F.linear(xhat, bk["wkv"].t[0], bk["bkv"]), kv_ref, atol=2e-5)
    assert torch.allclose(F.linear(xhat, bk["wq"].t[0], bk["bq"]), q_ref, atol=2e-5)
    idx = O.rel_pos_index(5).view(-1)
    rb = synth_sd[p + "attn.relative_position_bias_table"][idx].view(25, 25, 8).permute(2, 0, 1)
    assert torch.equal(bk["relbias"], rb.contiguous())
    w = torch.randn(7, 3, 5, 5)
    assert torch.equal(pack.conv_w(w)[2 * 5 + 3], w[:, :, 2, 3])
    wt = torch.randn(3, 7, 3, 3)
    assert torch.equal(pack.convT_w(wt)[1 * 3 + 2], wt[:, :, 1, 2].t())


def test_harness_metrics_match_cpu_formulas():
    """The harness computes PSNR / SSIM on the device (float64 band-matrix Gaussian); same numbers as the numpy/scipy forms."""
    from speinet_amd import inference, selection
    rng = np.random.RandomState(3)
    for h, w in ((32, 40), (57, 33)):
        a = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        b = np.clip(a.astype(int) + rng.randint(-25, 25, a.shape), 0, 255).astype(np.uint8)
        p, s = inference.metrics_gpu(torch.from_numpy(a), torch.from_numpy(b))
        assert abs(p.item() - selection.calc_psnr(a, b)) < 1e-9 and abs(s.item() - inference.calc_ssim(a, b)) < 1e-12
    p, _ = inference.metrics_gpu(torch.from_numpy(a), torch.from_numpy(a))
    assert p.item() == float("inf")


def test_checkpoint_tooling(tmp_path, synth_sd):
    """Export -> check -> strict load round trip; DataParallel `module.` prefix; derived buffers optional; mismatches reported."""
    from speinet_amd import checkpoint
    net = SPEINet(args=default_args())
    net.load_state_dict(synth_sd, strict=True)
    path = str(tmp_path / "model_best.pt")
    checkpoint.export(net, path)
    assert checkpoint.main(["check", path]) == 0
    other = SPEINet(args=default_args())
    checkpoint.load_into(other, path)
    assert all(torch.equal(v, other.state_dict()[k]) for k, v in net.state_dict().items())
    sd = {"module." + k: v for k, v in checkpoint.read(path).items() if not k.endswith(checkpoint.DERIVED)}
    torch.save(sd, path)
    checkpoint.load_into(SPEINet(args=default_args()), path)                 # prefixed, without the derived buffers
    sd.pop("module.fusion.weight")
    sd["module.recons_net.inBlock.0.0.weight"] = torch.zeros(32, 3, 3, 3)
    torch.save(sd, path)
    missing, unexpected, bad = checkpoint.validate(checkpoint.read(path))
    assert missing == ["fusion.weight"] and not unexpected and len(bad) == 1 and checkpoint.main(["check", path]) == 1
    with pytest.raises(RuntimeError):
        checkpoint.load_into(SPEINet(args=default_args()), path)


def test_encoder_cache_lru():
    """EncoderCache: LRU eviction, hit / miss counters (the values are opaque to it)."""
    from speinet_amd.speinet import EncoderCache
    c = EncoderCache(capacity=3)
    assert c.get("a") is None and c.misses == 1
    for k in "abc":
        c.put(k, k.upper())
    assert c.get("a") == "A"                      # refreshes "a"
    c.put("d", "D")                               # evicts the least recently used entry: "b"
    assert c.get("b") is None and c.get("a") == "A" and c.get("c") == "C" and c.get("d") == "D"
    assert (c.hits, c.misses) == (4, 2)
    c.clear()
    assert c.get("a") is None


def test_loss_l1_hem_vs_reference_golden(golden_dir):
    """speinet_amd.loss against the reference's own loss values (G20: nn.L1Loss + its Loss/hard_example_mining.py on the
    reference's output, numpy's global generator seeded as in the fixture): same masks, same numbers."""
    import numpy as np
    import torch
    from speinet_amd.loss import Loss
    from speinet_amd.synth import synth_frames
    for name in ("g20_train_swint_40x40", "g20_train_swint_n1_40x60"):
        d = np.load(os.path.join(golden_dir, name + ".npz"))
        seed, b, h, w = (int(d[k]) for k in ("seed", "b", "h", "w"))
        out = torch.from_numpy(d["out"]).requires_grad_(True)
        gt = synth_frames(b, h, w, seed=seed + 500)[:, 1].contiguous()
        np.random.seed(seed)
        fn = Loss("1*L1+2*HEM", device="cpu")
        loss = fn(out, gt)
        assert abs(loss.item() - float(d["loss"])) < 1e-7
        assert abs(fn.log[-1][0] - float(d["l1"])) < 1e-7 and abs(fn.log[-1][1] - 2 * float(d["hem"])) < 1e-7
        loss.backward()
        assert out.grad.abs().sum().item() > 0


def test_drop_path_rates_and_stream():
    """DropPath: the decay rule of model/swinir.py:691 and the draw order / algorithm recorded by the reference run (G20)."""
    import numpy as np
    import torch
    from speinet_amd.train import drop_path_rates, drop_path_scales
    r = drop_path_rates([6] * 6)
    assert len(r) == 36 and r[0] == 0.0 and abs(r[-1] - 0.1) < 1e-7 and all(b > a for a, b in zip(r, r[1:]))
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g20_train_swint_40x40.npz"))
    torch.manual_seed(int(d["seed"]))
    sc = drop_path_scales([6] * 6, int(d["b"]), 2)
    flat = [t for call in sc for pair in call if pair is not None for t in pair]
    assert len(flat) == d["draws"].shape[0] == 2 * 35 * 2
    assert all(torch.equal(a, torch.from_numpy(b)) for a, b in zip(flat, d["draws"]))
