"""GPU suite: the inference harness end to end on two synthetic clips (BASELINE.json configs[0] shape: frames on disk,
label files, 5-frame windows) — outputs identical to calling the model directly, log lines in the reference format,
and the detector path when the label file is missing."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from speinet_amd import inference, selection          # noqa: E402
from speinet_amd.synth import synth_frames            # noqa: E402


def _make_clip(root, clip, n, h, w, seed, labels=None):
    from PIL import Image
    x = synth_frames(1, h, w, seed=seed)[0]
    for sub in ("blur", "gt"):
        os.makedirs(os.path.join(root, sub, clip), exist_ok=True)
    for i in range(n):
        fr = torch.roll(x[i % 5], shifts=(i, -i), dims=(1, 2))
        img = (fr.permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "blur", clip, f"{i:06d}.png"))
        Image.fromarray(img).save(os.path.join(root, "gt", clip, f"{i:06d}.png"))
    if labels is not None:
        os.makedirs(os.path.join(root, "label"), exist_ok=True)
        np.save(os.path.join(root, "label", clip + ".npy"), np.asarray(labels))


def test_harness_two_clips(tmp_path):
    root, res = str(tmp_path / "data"), str(tmp_path / "res")
    _make_clip(root, "clipA", 6, 40, 60, 1, labels=[1, 0, 0, 0, 1, 0])
    _make_clip(root, "clipB", 5, 40, 60, 2, labels=None)            # no label file -> LD detector
    a = inference.build_args(["--data_path", root, "--model_path", "synthetic", "--result_path", res, "--precision", "f32"])
    inf = inference.Inference(a)
    tot = inf.infer()
    assert int(tot[2].item()) == 11
    logs = [f for f in os.listdir(res) if f.startswith("inference_log")]
    text = open(os.path.join(res, logs[0])).read()
    assert len(re.findall(r"^> clip[AB]-\d{6} PSNR=[\d.]+, SSIM=[\d.]+ pre_time:", text, flags=re.M)) == 11
    assert "# Total AVG-PSNR=" in text and "# Video:clipA AVG-PSNR=" in text
    # one frame recomputed by hand through the public model API
    blur = sorted(os.path.join(root, "blur", "clipA", f) for f in os.listdir(os.path.join(root, "blur", "clipA")))
    w = selection.assemble_windows(blur, np.load(os.path.join(root, "label", "clipA.npy")))[2]
    imgs = [inference._imread(p) for p in w["window"] + [w["pre"], w["sub"]]]
    if w["zero_pre"]:
        imgs[-2] = np.zeros_like(imgs[-2])
    if w["zero_sub"]:
        imgs[-1] = np.zeros_like(imgs[-1])
    assert torch.equal(selection.numpy2tensor_device(imgs, "cuda:0").cpu(), selection.numpy2tensor(imgs))
    with torch.no_grad():
        out = inf.net(selection.numpy2tensor(imgs).cuda())
    saved = inference._imread(os.path.join(res, "clipA", w["name"] + ".png"))
    assert np.array_equal(selection.tensor2numpy(out), saved)


def test_harness_frames_vs_oracle(tmp_path, synth_sd):
    """The harness end to end against the ORACLE, not against itself: the PNGs it writes for a clip (label file, LD selection, window
    assembly, zeroed far references, forward, uint8 conversion) against frames computed by the CPU restatement of the reference from
    the same files through the reference's own conversions (selection.numpy2tensor / tensor2numpy, G12): every pixel within one grey
    level, >= 99.9 % identical, and the PSNR it logs within 0.01 dB of the oracle frame's."""
    from oracle import speinet_oracle as O
    root, res = str(tmp_path / "data"), str(tmp_path / "res")
    _make_clip(root, "clipA", 6, 40, 60, 3, labels=[0, 1, 0, 0, 0, 1])
    a = inference.build_args(["--data_path", root, "--model_path", "synthetic", "--result_path", res, "--precision", "f32"])
    inf = inference.Inference(a)
    inf.infer()
    text = open(os.path.join(res, [f for f in os.listdir(res) if f.startswith("inference_log")][0])).read()
    blur = sorted(os.path.join(root, "blur", "clipA", f) for f in os.listdir(os.path.join(root, "blur", "clipA")))
    gts = sorted(os.path.join(root, "gt", "clipA", f) for f in os.listdir(os.path.join(root, "gt", "clipA")))
    wins = selection.assemble_windows(blur, np.load(os.path.join(root, "label", "clipA.npy")))
    gt_seqs, _ = selection.gene_seq(gts, 3, True)
    for k in (0, 2, 5):
        w = wins[k]
        imgs = [inference._imread(p) for p in w["window"] + [w["pre"], w["sub"]]]
        if w["zero_pre"]:
            imgs[-2] = np.zeros_like(imgs[-2])
        if w["zero_sub"]:
            imgs[-1] = np.zeros_like(imgs[-1])
        with torch.no_grad():
            ref = O.forward(selection.numpy2tensor(imgs), synth_sd, O.Cfg())
        ref_u8 = selection.tensor2numpy(ref)
        saved = inference._imread(os.path.join(res, "clipA", w["name"] + ".png"))
        diff = np.abs(saved.astype(np.int32) - ref_u8.astype(np.int32))
        assert diff.max() <= 1 and (diff == 0).mean() >= 0.999, (k, diff.max(), (diff == 0).mean())
        gt = inference._imread(gt_seqs[k][1])
        psnr_ref = O.psnr_uint8(torch.from_numpy(ref_u8), torch.from_numpy(gt))
        logged = float(re.search(rf"^> clipA-{w['name']} PSNR=([\d.]+)", text, flags=re.M).group(1))
        assert abs(logged - psnr_ref) < 1e-2 + 5e-5 * psnr_ref, (k, logged, psnr_ref)     # the log prints 5 significant digits


def test_harness_recomputes_non_finite_frames(tmp_path):
    """Half operands do not saturate: a frame with a non-finite value (here: one weight beyond +-65504, infinite as a half) is
    recomputed in split-bf16 arithmetic and counted, the clip is not aborted."""
    root, res = str(tmp_path / "data"), str(tmp_path / "res")
    _make_clip(root, "clipA", 4, 40, 60, 4, labels=[1, 0, 0, 1])
    a = inference.build_args(["--data_path", root, "--model_path", "synthetic", "--result_path", res, "--precision", "f16"])
    inf = inference.Inference(a)
    with torch.no_grad():
        inf.net.recons_net.outBlock[3].weight[0, 0, 0, 0] = 1.0e5
    inf.net.invalidate_packed()
    tot = inf.infer()
    assert int(tot[2].item()) == 4 and inf.range_retries == 4
    text = open(os.path.join(res, [f for f in os.listdir(res) if f.startswith("inference_log")][0])).read()
    assert text.count("recomputed in bf16x3") == 4 and np.isfinite(tot[0].item())


@pytest.mark.parametrize("graph", [False, True])
def test_forward_window_reuse_bit_identical(graph):
    """Cross-window reuse of the per-frame encoder passes (SURVEY.md plan step 8): sliding windows over a clip through
    `forward_window` give exactly the bits of the stateless `forward`, with most encoder passes served from the cache;
    both routing branches, a zeroed reference frame, eager and hipGraph tails."""
    from speinet_amd.speinet import EncoderCache, SPEINet, default_args
    from speinet_amd.synth import state_dict_template, synth_state_dict
    net = SPEINet(args=default_args())
    net.load_state_dict(synth_state_dict(state_dict_template(), seed=0))
    net = net.cuda().eval()
    net.precision, net.corr_precision, net.streams, net.use_graph = "bf16", "bf16", 2, graph
    frames = synth_frames(2, 60, 80, seed=11).reshape(10, 3, 60, 80)[:8].cuda()       # an 8-frame "clip"
    zero = torch.zeros_like(frames[0])
    cache = EncoderCache()
    with torch.no_grad():
        for t in range(1, 7):
            pre_zero, sub_zero = t == 3, t == 5                                          # window 3 -> SelfTransfer branch
            pre, sub = (zero if pre_zero else frames[0]), (zero if sub_zero else frames[7])
            x = torch.stack([frames[t - 1], frames[t], frames[t + 1], pre, sub]).unsqueeze(0)
            keys = [t - 1, t, t + 1, "zero" if pre_zero else 0, "zero" if sub_zero else 7]
            out_w = net.forward_window(x, keys, cache, zero_ref=pre_zero)
            net.use_graph = False
            ref = net(x, routing=[pre_zero])
            net.use_graph = graph
            assert torch.equal(out_w, ref), f"window {t}"
    # 8 raw + 6 RL-5 + 8 RL-1 encoder passes and 2 reference pyramids were computed, instead of 36 + 5 without the cache
    assert cache.misses == 24 and cache.hits >= 10, (cache.hits, cache.misses)


@pytest.mark.parametrize("h,w", [(720, 1280), (60, 100), (45, 77), (19, 51)])
def test_frame_post_vs_numpy(h, w):
    """csrc/metrics.hip (uint8 conversion + finite flag + PSNR + SSIM in three launches) against the host restatements of the
    reference's functions: `selection.tensor2numpy` (inference_SPEINet.py:477-482), `selection.calc_psnr` (:484-500) and
    `inference.calc_ssim` (:502-543, numpy float64; SSIM itself is parity-unpinned: cv2 is absent) on the 4-pixel-cropped frames."""
    from speinet_amd import ops
    g = torch.Generator().manual_seed(h * 1000 + w)
    gt = (torch.rand(h, w, 3, generator=g) * 255).round().to(torch.uint8)
    out = (gt.permute(2, 0, 1).float() / 255 + 0.08 * torch.randn(3, h, w, generator=g)).contiguous()     # leaves [0, 1] in places
    out[0, 5, 7] = 0.5 / 255                                                                               # a tie: rounds to even (0)
    out[1, 6, 8] = 1.5 / 255                                                                               # ... (2)
    u8, res = ops.frame_post(out.to("cuda:0"), gt.to("cuda:0"), 4)
    ref_u8 = selection.tensor2numpy(out[None])
    assert np.array_equal(u8.cpu().numpy(), ref_u8)
    fin, psnr, ssim = res.tolist()
    assert fin == 1.0
    a, b = ref_u8[4:-4, 4:-4], gt.numpy()[4:-4, 4:-4]
    assert abs(psnr - selection.calc_psnr(a, b)) < 1e-9
    assert abs(ssim - inference.calc_ssim(a, b)) < 1e-10
    # identical frames: PSNR inf, SSIM 1; a non-finite value is reported and does not poison the uint8 frame
    same = (gt.permute(2, 0, 1).float() / 255).contiguous()
    _, res = ops.frame_post(same.to("cuda:0"), gt.to("cuda:0"), 4)
    assert res[0].item() == 1.0 and res[1].item() == float("inf") and abs(res[2].item() - 1.0) < 1e-12
    same[2, 3, 3] = float("nan")
    same[0, 9, 9] = float("inf")
    u8, res = ops.frame_post(same.to("cuda:0"), gt.to("cuda:0"), 4)
    assert res[0].item() == 0.0 and u8.cpu()[3, 3, 2] == 0
