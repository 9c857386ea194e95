"""GPU suite: LD-detector focus measures (row a11) through the C-ABI vs the reference's own outputs (golden G13, five
measures) and vs the oracle (all six, incl. the parity-unpinned WAV1 and the gray conversion).  Tolerance 2e-4 relative
(fp32 sums over 5k..900k terms in a different order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detector_oracle as D      # noqa: E402
from speinet_amd import detector              # noqa: E402


def test_features_vs_reference_golden(golden_dir):
    g13 = np.load(os.path.join(golden_dir, "g13_detector.npz"))
    g = torch.from_numpy(g13["gray"])[:, 0].cuda()
    for k in (11, 7):
        f = detector.gray_focus_measures(g, k).cpu().numpy()
        for col, name in ((0, "lap1"), (1, "mis3"), (3, "gra7"), (4, "sta3"), (5, "dct3")):
            np.testing.assert_allclose(f[:, col], g13[f"{name}_k{k}"], rtol=2e-4, err_msg=f"{name} k={k}")


@pytest.mark.parametrize("h,w", [(64, 80), (97, 131), (200, 320)])
def test_features_vs_oracle(h, w):
    r = np.random.RandomState(h)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    frames = np.stack([np.clip(128 + 90 * np.sin(0.05 * (i + 1) * yy) * np.cos(0.08 * xx) + (3 + 10 * i) * r.randn(3, h, w), 0, 255)
                       for i in range(3)]).astype(np.float32)
    t = torch.from_numpy(frames)
    ref = D.features(t, 11)
    out = detector.focus_measures(t.cuda(), 11).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-4)
    assert np.array_equal(detector.predict(out), D.predict(ref).numpy())
