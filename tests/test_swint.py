"""The `swint` variant (reference model/swint.py, `--model swint`): oracle and HIP path against the reference's own outputs
(tests/golden/make_golden_swint.py, G18) and the state_dict inventory the reference module wrote."""
import os

import numpy as np
import pytest
import torch

from oracle import speinet_oracle as O
from speinet_amd.synth import synth_frames, synth_state_dict

CASES = [("g18_swint_40x60", 3, 2, 40, 60), ("g18_swint_100x100", 3, 1, 100, 100), ("g18_swint_n1_40x60", 1, 1, 40, 60)]


def _template(golden_dir, n_seq):
    sd = {}
    for line in open(os.path.join(golden_dir, "state_dict_keys_swint.txt")):
        k, shp, dt = line.rstrip("\n").split("\t")
        shape = tuple(int(x) for x in shp.split(",")) if shp else ()
        if k == "conv.weight":
            shape = (128, 128 * n_seq, 1, 1)
        sd[k] = torch.zeros(shape, dtype=getattr(torch, dt))
    from speinet_amd.synth import _rel_pos_index, _shift_mask
    for k in sd:
        if k.endswith("relative_position_index"):
            sd[k] = _rel_pos_index(5)
        elif k.endswith("attn_mask"):
            sd[k] = _shift_mask(50, 50, 5, 2)
    return sd


def _case(golden_dir, name, n_seq, b, h, w):
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    x = synth_frames(b, h, w, seed=int(d["seed"]))[:, :n_seq]
    return torch.from_numpy(d["out"]), x, synth_state_dict(_template(golden_dir, n_seq), seed=0)


def test_state_dict_inventory(golden_dir):
    """Same 990 names, shapes and dtypes as the reference's model/swint.py module."""
    from speinet_amd.swint import SPEINet
    from speinet_amd.speinet import default_args
    ours = SPEINet(args=default_args()).state_dict()
    ref = _template(golden_dir, 3)
    assert list(ours.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(ours[k].shape) == tuple(ref[k].shape) and ours[k].dtype == ref[k].dtype, k
    a = default_args()
    a.n_sequence = 1
    from speinet_amd.swint import make_model
    assert make_model(a).conv.weight.shape == (128, 128, 1, 1)


@pytest.mark.parametrize("name,n_seq,b,h,w", CASES)
def test_oracle_vs_reference(golden_dir, name, n_seq, b, h, w):
    ref, x, sd = _case(golden_dir, name, n_seq, b, h, w)
    cfg = O.Cfg(n_sequence=n_seq)
    with torch.no_grad():
        out = O.forward_swint(x, sd, cfg)
    assert (out - ref).abs().max().item() < 2e-5


def test_no_cpu_path():
    from speinet_amd.swint import SPEINet
    with pytest.raises(RuntimeError):
        SPEINet().eval()(torch.zeros(1, 3, 3, 20, 20))


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_seq,b,h,w", CASES)
def test_hip_vs_reference(golden_dir, name, n_seq, b, h, w):
    """Every arithmetic mode against the reference's output: f32-grade modes 1e-3 absolute / 1e-3 dB, f16 (the throughput
    mode) 3e-3 / 1e-3 dB, bf16 its documented bound; 2 HIP streams give the bits of 1."""
    from speinet_amd.swint import SPEINet
    from speinet_amd.speinet import default_args
    ref, x, sd = _case(golden_dir, name, n_seq, b, h, w)
    net = SPEINet(n_sequence=n_seq, args=default_args())
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").eval()
    xd = x.to("cuda:0")
    outs = {}
    for mode, tol, tol_db in (("f32", 1e-3, 1e-3), ("bf16x3", 1e-3, 1e-3), ("f16", 3e-3, 1e-3), ("bf16", 0.05, 1e-2)):
        net.precision = mode
        with torch.no_grad():
            out = net(xd).cpu()
        outs[mode] = out
        err = (out - ref).abs().max().item()
        dp = max(abs(O.psnr_uint8(O.to_uint8(out[i:i + 1]), O.to_uint8(x[i:i + 1, n_seq // 2])) -
                     O.psnr_uint8(O.to_uint8(ref[i:i + 1]), O.to_uint8(x[i:i + 1, n_seq // 2]))) for i in range(b))
        print(f"{name} {mode}: max abs err {err:.2e}, |dPSNR| {dp:.1e} dB")
        assert err < tol and dp <= tol_db, mode
    net.precision, net.streams = "f16", 2
    with torch.no_grad():
        assert torch.equal(net(xd).cpu(), outs["f16"])
