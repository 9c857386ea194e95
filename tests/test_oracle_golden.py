"""CPU suite: the oracle restatement vs golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Tolerances are fp32 round-off of re-ordered sums (oneDNN vs our
restatement use the same ATen ops, so most cases agree to ~1e-6)."""
import os

import numpy as np
import pytest
import torch

from oracle import speinet_oracle as O
from speinet_amd.synth import synth_frames

torch.set_num_threads(8)


def g(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: (torch.from_numpy(d[k]) if d[k].ndim > 0 else d[k].item()) for k in d.files}


def close(a, b, rtol=1e-4, atol=1e-5):
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"max abs err {err:.3e} vs ref max {ref:.3e}"


def test_state_dict_inventory(synth_sd):
    assert len(synth_sd) == 1020
    n = sum(v.numel() for k, v in synth_sd.items() if v.dtype.is_floating_point and not k.endswith("attn_mask")
            and not k.endswith("running_mean") and not k.endswith("running_var"))
    assert n == 30_830_403 - 0 or n > 30_000_000


def test_g01_rl_prior(golden_dir):
    d = g(golden_dir, "g01_rl")
    close(O.rl_prior(d["x"], 1), d["it1"], 1e-5, 1e-6)
    close(O.rl_prior(d["x"], 5), d["it5"], 1e-5, 1e-6)
    assert not torch.isnan(d["it5"]).any()


@pytest.mark.parametrize("c", [32, 64, 128])
def test_g02_resblock(golden_dir, synth_sd, c):
    d = g(golden_dir, f"g02_resblock{c}")
    close(O.resblock(d["x"], synth_sd, str(d["key"])), d["out"])


def test_g03_encoder(golden_dir, synth_sd):
    d = g(golden_dir, "g03_enc")
    cfg = O.Cfg()
    lv1 = O.in_block(d["x"], synth_sd, cfg)
    lv2 = O.encoder_first(lv1, synth_sd, cfg)
    lv3 = O.encoder_second(lv2, synth_sd, cfg)
    close(lv1, d["lv1"]); close(lv2, d["lv2"]); close(lv3, d["lv3"])


def test_g04_window_attention(golden_dir, synth_sd):
    d = g(golden_dir, "g04_winattn")
    p0 = "swin.layers.0.residual_group.blocks.0.attn."
    p1 = "swin.layers.0.residual_group.blocks.1.attn."
    close(O.window_attention(d["xw"], d["yw"], synth_sd, p0, 8, 5, None), d["out_nomask"])
    close(O.window_attention(d["xw"], d["yw"], synth_sd, p1, 8, 5, d["mask"]), d["out_mask"])
    assert torch.equal(O.shift_mask(10, 15, 5, 2), d["mask"])


def test_g05_swin_block(golden_dir, synth_sd):
    d = g(golden_dir, "g05_block_10x15")
    p = "swin.layers.0.residual_group.blocks."
    close(O.swin_block(d["xt"], d["yt"], (10, 15), synth_sd, p + "0.", 8, 5, 0), d["out_s0"])
    close(O.swin_block(d["xt"], d["yt"], (10, 15), synth_sd, p + "1.", 8, 5, 2), d["out_s2"])
    d = g(golden_dir, "g05_block_50x50")
    rnd = lambda s, *sh: torch.from_numpy(np.random.RandomState(s).randn(*sh).astype(np.float32))
    xt, yt = rnd(d["seed_x"], 1, 2500, 256), rnd(d["seed_y"], 1, 2500, 256)
    close(O.swin_block(xt, yt, (50, 50), synth_sd, p + "0.", 8, 5, 0)[:, ::7], d["out_s0_sub"])
    close(O.swin_block(xt, yt, (50, 50), synth_sd, p + "1.", 8, 5, 2)[:, ::7], d["out_s2_sub"])


def test_attn_mask_buffer_matches_formula(synth_sd):
    # the registered [100,25,25] buffer (patch 200 -> 50x50) equals calculate_mask((50,50))
    assert torch.equal(synth_sd["swin.layers.0.residual_group.blocks.1.attn_mask"], O.shift_mask(50, 50, 5, 2))
    assert torch.equal(synth_sd["swin.layers.0.residual_group.blocks.0.attn.relative_position_index"], O.rel_pos_index(5))


def test_g06_swin(golden_dir, synth_sd):
    d = g(golden_dir, "g06_swin")
    close(O.swin(d["x"], d["y"], synth_sd, O.Cfg()), d["out"])


@pytest.mark.parametrize("name", ["g07_search", "g07_search_tie"])
def test_g07_search_transfer(golden_dir, name):
    d = g(golden_dir, name)
    s, t3, t2, t1, arg = O.search_transfer(d["lr3"], d["rf3"], d["rf1"], d["rf2"], d["rf3"], return_arg=True)
    assert torch.equal(arg, d["arg"])          # index work: bit exact, first-max tie-break
    close(s, d["s"]); close(t3, d["t3"]); close(t2, d["t2"]); close(t1, d["t1"])


def test_g08_self_transfer(golden_dir, synth_sd):
    d = g(golden_dir, "g08_self")
    s, t3, t2, t1 = O.self_transfer(d["x"], synth_sd)
    close(s, d["s"]); close(t3, d["t3"]); close(t2, d["t2"]); close(t1, d["t1"])


def test_g09_decode(golden_dir, synth_sd):
    d = g(golden_dir, "g09_decode")
    close(O.decode(d["ff"], d["s"], d["t3"], d["t2"], d["t1"], synth_sd, O.Cfg()), d["out"])


@pytest.mark.parametrize("name,b,h,w", [("g10_fwd_40x60_mixed", 2, 40, 60), ("g10_fwd_100x100", 1, 100, 100)])
def test_g10_forward(golden_dir, synth_sd, name, b, h, w):
    d = g(golden_dir, name)
    zr = tuple(int(i) for i in d["zero_ref"]) if "zero_ref" in d else ()
    x = synth_frames(b, h, w, seed=int(d["seed"]), zero_ref=zr)
    out = O.forward(x, synth_sd, O.Cfg())
    close(out, d["out"], 2e-4, 1e-5)


def test_g16_forward_480x640_subsampled(golden_dir, synth_sd):
    """The oracle at the BSD frame size (mixed-routing batch of two; about a minute on 8 threads) against the reference's own
    output, of which G16 keeps every 8th pixel and the per-channel statistics."""
    d = np.load(os.path.join(golden_dir, "g16_fwd_480x640_mixed.npz"))
    x = synth_frames(2, 480, 640, seed=int(d["seed"]), zero_ref=tuple(int(i) for i in d["zero_ref"]))
    with torch.no_grad():
        out = O.forward(x, synth_sd)
    close(out[:, :, ::8, ::8], torch.from_numpy(d["sub"]), 1e-3, 1e-4)
    assert (out.mean(dim=(2, 3)) - torch.from_numpy(d["mean"])).abs().max().item() < 1e-5
