"""CPU suite: host-side selection logic against golden vectors produced by the reference's own
Inference methods (tests/golden/make_golden_selection.py)."""
import json
import os

import numpy as np
import pytest
import torch

from speinet_amd import selection as S


@pytest.fixture(scope="module")
def g11(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g11_selection.json")))


def test_blurry_indices_all_patterns(g11):
    assert len(g11) >= 10
    for name, d in g11.items():
        pre, sub = S.blurry_indices(d["labels"])
        assert pre == d["pre"], name
        assert sub == d["sub"], name


@pytest.mark.parametrize("border", [True, False])
def test_window_generation(g11, border):
    for name, d in g11.items():
        frames = [f"clip/{i:06d}.png" for i in range(len(d["labels"]))]
        ref = d[f"border_{border}"]
        seqs, padded = S.gene_seq(frames, 3, border)
        assert seqs == ref["seqs"] and padded == ref["padded"], name
        pre_w, sub_w = S.gene_seq_nsf(np.array(d["labels"]), 3, border)
        assert pre_w == ref["pre_windows"] and sub_w == ref["sub_windows"], name


def test_assemble_windows_zeroing_rule(g11):
    d = g11["long_gaps"]
    frames = [f"clip/{i:06d}.png" for i in range(len(d["labels"]))]
    wins = S.assemble_windows(frames, np.array(d["labels"]))
    assert len(wins) == len(frames)
    padded = d["border_True"]["padded"]
    for w, pw, sw in zip(wins, d["border_True"]["pre_windows"], d["border_True"]["sub_windows"]):
        assert w["pre"] == padded[pw[0]] and w["sub"] == padded[sw[2]]
        last = S.frame_number(w["window"][2])                    # reference compares against frame_numbers[2]
        assert w["zero_pre"] == (abs(last - S.frame_number(w["pre"])) > 7)
    assert any(w["zero_pre"] for w in wins) and not all(w["zero_pre"] for w in wins)


def test_convert_and_psnr(golden_dir):
    d = np.load(os.path.join(golden_dir, "g12_convert.npz"))
    t = S.numpy2tensor(list(d["imgs"]))
    assert torch.equal(t, torch.from_numpy(d["tensor"]))
    u8 = S.tensor2numpy(torch.from_numpy(d["o"]))
    assert np.array_equal(u8, d["u8"])
    assert abs(S.calc_psnr(d["gt"][4:-4, 4:-4], u8[4:-4, 4:-4]) - float(d["psnr"])) < 1e-12
