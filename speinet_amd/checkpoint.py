"""Checkpoint tooling: validate a reference checkpoint against the 1020-entry inventory, load it
into the MI355X module, and write reference-compatible state_dicts back.

The module keeps the reference's parameter names and layouts (`speinet_amd/speinet.py`), so `state_dict()` IS the
reference format; the packed kernel operands (`speinet_amd/pack.py`) are derived data and never serialised.
Reference call sites: `inference_SPEINet.py:230-233` (strict `load_state_dict(torch.load(model_path))`),
`model/__init__.py:40-82` (`save` / `load`, optional `module.` prefix under DataParallel, `strict=False` resume).
Files are read with `torch.load(..., weights_only=True)` only: nothing in a checkpoint is executed.

    python -m speinet_amd.checkpoint check  <model_best.pt>
    python -m speinet_amd.checkpoint export <out.pt> [--synthetic-seed N]
"""
from __future__ import annotations

import argparse
import sys
from typing import Dict, List, Tuple

import torch

from .synth import state_dict_template, synth_state_dict

# buffers the reference rebuilds in its constructor; a checkpoint may carry them or not (swinir.py:102, :213)
DERIVED = ("relative_position_index", "attn_mask")


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """DataParallel checkpoints prefix every key with `module.` (model/__init__.py:33-37, 46)."""
    if sd and all(k.startswith("module.") for k in sd):
        return {k[len("module."):]: v for k, v in sd.items()}
    return sd


def validate(sd: Dict[str, torch.Tensor]) -> Tuple[List[str], List[str], List[str]]:
    """(missing, unexpected, mismatched) against the reference inventory (names, shapes, dtypes)."""
    ref = state_dict_template()
    sd = strip_prefix(sd)
    missing = [k for k in ref if k not in sd and not k.endswith(DERIVED)]
    unexpected = [k for k in sd if k not in ref]
    bad = [f"{k}: {tuple(sd[k].shape)} {sd[k].dtype} != {tuple(ref[k].shape)} {ref[k].dtype}"
           for k in ref if k in sd and (tuple(sd[k].shape) != tuple(ref[k].shape) or sd[k].dtype != ref[k].dtype)]
    return missing, unexpected, bad


def read(path: str) -> Dict[str, torch.Tensor]:
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f"{path}: not a state_dict (a name -> tensor mapping)")
    return strip_prefix(obj)


def load_into(model: torch.nn.Module, path: str, strict: bool = True):
    """`model.load_state_dict(torch.load(path))` with the checks spelled out; derived buffers may be absent."""
    sd = read(path)
    missing, unexpected, bad = validate(sd)
    if bad or (strict and (missing or unexpected)):
        raise RuntimeError(f"{path}: {len(missing)} missing, {len(unexpected)} unexpected, {len(bad)} mismatched entries; "
                           f"first: {(bad + missing + unexpected)[0]}")
    own = model.state_dict()
    for k in own:
        if k not in sd and k.endswith(DERIVED):
            sd[k] = own[k]
    return model.load_state_dict(sd, strict=strict)


def export(model: torch.nn.Module, path: str) -> None:
    """Write the module's parameters in the reference layout (what `Model.save` writes, model/__init__.py:40-53)."""
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    missing, unexpected, bad = validate(sd)
    assert not (missing or unexpected or bad), (missing[:1], unexpected[:1], bad[:1])
    torch.save(sd, path)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("check")
    c.add_argument("path")
    e = sub.add_parser("export")
    e.add_argument("path")
    e.add_argument("--synthetic-seed", type=int, default=0)
    a = ap.parse_args(argv)
    if a.cmd == "check":
        sd = read(a.path)
        missing, unexpected, bad = validate(sd)
        n = sum(v.numel() for v in sd.values())
        print(f"{a.path}: {len(sd)} entries, {n} values; missing {len(missing)}, unexpected {len(unexpected)}, mismatched {len(bad)}")
        for tag, lst in (("missing", missing), ("unexpected", unexpected), ("mismatched", bad)):
            for k in lst[:10]:
                print(f"  {tag}: {k}")
        return 0 if not (missing or unexpected or bad) else 1
    from .speinet import SPEINet, default_args
    net = SPEINet(args=default_args())
    net.load_state_dict(synth_state_dict(state_dict_template(), seed=a.synthetic_seed), strict=True)
    export(net, a.path)
    print(f"wrote {a.path}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
