"""ORACLE — test infrastructure only.  CPU restatement (PyTorch fp32) of the reference's LD sharpness detector at
inference time: six focus measures per frame + logistic-regression predict.

Reference: inference_SPEINet.py:54-189 (sobel :54, laplacian :68, mask :79, focus_measure_mis3 :118, _gra7 :134,
_lap1 :144, _wave1 :152, _sta3 :161, _dct3 :169, generate_vars :177-189), predict :351-353.

Pinning: LAP1, MIS3, GRA7, STA3, DCT3 are checked against golden vectors produced by calling the reference's own
functions (tests/golden/make_golden_detector.py -> g13_detector.npz).  **Parity unpinned** (the packages are absent
here and the reference pins no versions): WAV1 (`ptwt.wavedec2` with `pywt.Wavelet('db6')`, mode 'zero') is restated
from the published Daubechies-6 filter bank, and the RGB->gray step (`torchvision.transforms.Grayscale`) uses the
ITU-R 601 weights 0.2989/0.587/0.114.  The logistic-regression weights are the 7 numbers of
LD_detector/pickle/LogisticRegression_0.5_11.pkl, read from the raw pickle bytes with pickletools (never unpickled) and
cross-checked against LD_detector/output.csv:158.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# LogisticRegression_0.5_11: coef_ (LAP1, MIS3, WAV1, GRA7, STA3, DCT3), intercept_
LR_COEF = (-0.11971818612047416, -1.2293425023576632, 0.0044214112366378735, -0.042858891031731176,
           0.12867998448379486, 1.5577974574202265)
LR_INTERCEPT = -1.5940041517368388

# Daubechies-6 (12 taps) decomposition low-pass filter, pywt convention (dec_lo); dec_hi[k] = (-1)^(k+1) dec_lo[11-k]
DB6_DEC_LO = (-0.00107730108499558, 0.004777257511010651, 0.0005538422009938016, -0.031582039318031156,
              0.02752286553001629, 0.09750160558707936, -0.12976686756709563, -0.22626469396516913,
              0.3152503517092432, 0.7511339080215775, 0.4946238903983854, 0.11154074335008017)


def gray(frames: torch.Tensor) -> torch.Tensor:
    """[N,3,H,W] (0..255) -> [N,1,H,W] in 0..1  (generate_vars :182; torchvision Grayscale weights: unpinned)."""
    r, g, b = frames[:, 0:1], frames[:, 1:2], frames[:, 2:3]
    return (0.2989 * r + 0.587 * g + 0.114 * b) / 255.0


def _lp_sq_mean(x: torch.Tensor, k: int) -> torch.Tensor:
    """mean over windows of lp_pool2d(x, norm 2, k)**2 == mean over windows of sum(x^2)."""
    return (F.lp_pool2d(x, norm_type=2, kernel_size=k) ** 2).mean(dim=(1, 2, 3))


def lap1(g: torch.Tensor, k: int) -> torch.Tensor:
    w = torch.tensor([[1, 1, 1], [1, -8, 1], [1, 1, 1]], dtype=torch.float32).view(1, 1, 3, 3)
    return _lp_sq_mean(F.conv2d(g, w, padding=1), k)


def mis3(g: torch.Tensor, k: int) -> torch.Tensor:
    f = torch.zeros(9, 1, 3, 3)
    f[:, :, 1, 1] = 1
    for i, (a, b) in enumerate(((0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2))):
        f[i, 0, a, b] = 0 if i == 4 else -1
    c = F.conv2d(g, f, padding=1).abs().sum(dim=1, keepdim=True)
    return F.lp_pool2d(c, norm_type=1, kernel_size=k).mean(dim=(1, 2, 3))


def sobel(g: torch.Tensor) -> torch.Tensor:
    gx = torch.tensor([[1.0, 0.0, -1.0], [2.0, 0.0, -2.0], [1.0, 0.0, -1.0]])
    gy = torch.tensor([[1.0, 2.0, 1.0], [0.0, 0.0, 0.0], [-1.0, -2.0, -1.0]])
    x = F.conv2d(g, torch.stack((gx, gy)).unsqueeze(1), padding=1)
    return (x ** 2).sum(dim=1, keepdim=True).sqrt()


def gra7(g: torch.Tensor, k: int) -> torch.Tensor:
    s = sobel(g)
    return _lp_sq_mean(s - F.avg_pool2d(s, kernel_size=k, padding=k // 2, stride=1), k)


def sta3(g: torch.Tensor, k: int) -> torch.Tensor:
    return _lp_sq_mean(g - F.avg_pool2d(g, kernel_size=k, padding=k // 2, stride=1), k)


def dct3(g: torch.Tensor, k: int) -> torch.Tensor:
    m = torch.tensor([[1, 1, -1, -1], [1, 1, -1, -1], [-1, -1, 1, 1], [-1, -1, 1, 1]], dtype=torch.float32).view(1, 1, 4, 4)
    return (F.lp_pool2d(F.conv2d(g, m), kernel_size=k, norm_type=1) ** 2).mean(dim=(1, 2, 3))


def wav1(g: torch.Tensor) -> torch.Tensor:
    """sum |LH| + |HL| + |HH| of a level-1 db6 DWT with zero extension (parity unpinned, see module docstring)."""
    lo = torch.tensor(DB6_DEC_LO, dtype=torch.float64)
    hi = torch.tensor([(-1) ** (i + 1) * DB6_DEC_LO[11 - i] for i in range(12)], dtype=torch.float64)
    x = g.double()

    def analysis(t, filt, dim):        # pywt 'zero' mode: full convolution, keep odd samples
        t = t.movedim(dim, -1)
        shp = t.shape
        y = F.conv1d(t.reshape(-1, 1, shp[-1]), filt.flip(0).view(1, 1, -1), padding=11)[..., 1::2]
        return y.reshape(*shp[:-1], -1).movedim(-1, dim)

    lo_w, hi_w = analysis(x, lo, 3), analysis(x, hi, 3)
    lh = analysis(lo_w, hi, 2)         # detail along rows of the column-low band
    hl = analysis(hi_w, lo, 2)
    hh = analysis(hi_w, hi, 2)
    return (lh.abs() + hl.abs() + hh.abs()).sum(dim=(1, 2, 3)).float()


def features(frames: torch.Tensor, k: int = 11) -> torch.Tensor:
    """[N,3,H,W] uint8-valued float frames -> [N,6] in the reference's order (LAP1, MIS3, WAV1, GRA7, STA3, DCT3)."""
    g = gray(frames.float())
    return torch.stack((lap1(g, k), mis3(g, k), wav1(g), gra7(g, k), sta3(g, k), dct3(g, k)), dim=1)


def predict(feat: torch.Tensor) -> torch.Tensor:
    """sklearn LogisticRegression.predict: 1 (sharp) iff w.f + b > 0."""
    w = torch.tensor(LR_COEF, dtype=torch.float64)
    return ((feat.double() @ w + LR_INTERCEPT) > 0).long()
