"""CPU suite: LD-detector oracle (row a11) vs golden vectors from the reference's own focus-measure functions."""
import os

import numpy as np
import pytest
import torch

from oracle import detector_oracle as D


@pytest.fixture(scope="module")
def g13(golden_dir):
    return np.load(os.path.join(golden_dir, "g13_detector.npz"))


@pytest.mark.parametrize("k", [11, 7])
def test_focus_measures_match_reference(g13, k):
    g = torch.from_numpy(g13["gray"])
    for name, fn in (("lap1", D.lap1), ("mis3", D.mis3), ("gra7", D.gra7), ("sta3", D.sta3), ("dct3", D.dct3)):
        ref = torch.from_numpy(g13[f"{name}_k{k}"])
        assert torch.allclose(fn(g, k), ref, rtol=1e-5, atol=1e-7), name


def test_wav1_properties():
    """WAV1 is parity-unpinned (ptwt absent): check the restated db6 bank instead — orthonormal filters, zero response
    to constants away from the border, energy growth with high-frequency content."""
    lo = torch.tensor(D.DB6_DEC_LO, dtype=torch.float64)
    assert abs(lo.sum().item() - 2 ** 0.5) < 1e-12 and abs((lo * lo).sum().item() - 1) < 1e-12
    for s in (2, 4):
        assert abs((lo[s:] * lo[:-s]).sum().item()) < 1e-12
    smooth = torch.full((1, 1, 64, 80), 0.5)
    noisy = smooth + 0.1 * torch.randn(1, 1, 64, 80, generator=torch.Generator().manual_seed(1))
    assert D.wav1(noisy).item() > 3 * D.wav1(smooth).item()


def test_lr_coefficients_match_output_csv():
    # LD_detector/output.csv:158 lists the same six coefficients (the intercept only lives in the pickle bytes)
    assert D.LR_COEF[1] == -1.2293425023576632 and D.LR_COEF[5] == 1.5577974574202265
    f = torch.tensor([[0.0] * 6, [0, 0, 0, 0, 0, 2.0]])
    assert D.predict(f).tolist() == [0, 1]
