"""ctypes loader for libspeinet_hip.so — the C-ABI declared in include/speinet_hip.h.

The product path has no CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspeinet_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "speinet_hip.h")

_lib = None

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float

# name -> (restype, argtypes); mirrors include/speinet_hip.h one to one
SIGNATURES = {
    "spei_version": (I, []),
    "spei_last_error": (C.c_char_p, []),
    "spei_arch": (C.c_char_p, []),
    "spei_any_nonzero": (I, [P, L, P, P]),
    "spei_rl_prior": (I, [P, P, P, I, I, I, I, F, P]),
    "spei_conv5_in": (I, [P, P, P, P, I, I, I, P]),
    "spei_conv5_out": (I, [P, I, P, P, P, I, I, I, P]),
    "spei_igemm_f32": (I, [P, I, I, P, I, I, P, P, P, I, P, I, P, I, I, I, I, I, I, I, I, I, I, P]),
    "spei_igemm_f32_batched": (I, [P, I, I, P, I, I, P, P, P, I, P, I, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "spei_igemm_bf16": (I, [P, I, I, P, I, I, P, P, P, P, I, P, I, P, I, I, I, I, I, I, I, I, I, I, P]),
    "spei_conv_slab16": (I, [I, P, I, I, P, I, I, I, P, P, P, P, I, I, P, I, P, I, I, I, I, I, I, I, I, I, I, P]),
    "spei_conv32_ws16": (I, [I, P, I, P, P, P, I, I, I, I, I, P]),
    "spei_conv_slab16_fa": (I, [I, P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "spei_attn_fused16": (I, [I, P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "spei_attn_win4_16": (I, [I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "spei_conv5_out_slab16": (I, [I, P, I, I, P, P, P, I, I, P]),
    "spei_convt2_slab16": (I, [I, P, I, I, I, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "spei_convt2_slab16x3": (I, [P, I, I, P, P, P, P, I, I, I, I, I, P]),
    "spei_conv3x3_256_pipe16": (I, [I, P, P, P, P, P, I, I, I, P]),
    "spei_mlp_fused16": (I, [I, P, P, P, P, P, P, L, P]),
    "spei_split16": (I, [I, P, I, P, P, L, I, P]),
    "spei_corr_slab16": (I, [I, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P]),
    "spei_corr_slab_top2_16": (I, [I, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P]),
    "spei_pack_split16": (I, [P, I, I, I, I, P, P, P]),
    "spei_corr_diag_ws_floats": (L, [I, I, I, I]),
    "spei_corr_diag_top2_16": (I, [I, P, P, P, I, I, I, I, I, P, P, P, P, P, P]),
    "spei_corr_rescore": (I, [P, I, P, I, P, P, I, I, I, I, I, P, P, P, P, P]),
    "spei_corr_argmax_bf16": (I, [P, P, P, P, P, P, I, I, I, I, I, P, P, P, P]),
    "spei_gate_ws_floats": (L, [I, I, I]),
    "spei_resblock_gates": (I, [P, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "spei_resblock_apply": (I, [P, P, I, P, P, P, P, P, I, I, I, I, P]),
    "spei_resblock_apply_batched": (I, [P, P, I, P, P, P, P, I, I, I, I, P]),
    "spei_resblock_gates_batched": (I, [P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "spei_conv_slab16_batched": (I, [I, P, I, I, P, P, P, P, I, P, I, I, I, I, I, I, I, I, I, I, P]),
    "spei_layernorm256": (I, [P, P, I, P, P, L, P]),
    "spei_window_attention": (I, [P, P, I, P, P, I, I, I, P]),
    "spei_window_attention_batched": (I, [P, P, I, P, P, I, I, I, I, P]),
    "spei_patch_invnorm": (I, [P, I, P, I, I, I, P]),
    "spei_corr_ws_floats": (L, [L]),
    "spei_corr_argmax": (I, [P, I, P, I, P, P, I, I, I, I, I, P, P, P, P]),
    "spei_gather_fold": (I, [P, I, P, P, I, I, I, I, I, I, I, P]),
    "spei_rot90": (I, [P, I, P, I, I, I, P]),
    "spei_upsample_bicubic": (I, [P, I, P, I, I, I, I, I, I, P]),
    "spei_add": (I, [P, P, P, L, P]),
    "spei_wgrad_ws_floats": (L, [I, I, I, I, I]),
    "spei_conv_wgrad_f32_batched": (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "spei_conv_wgrad_bf16x3_batched": (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "spei_conv_wgrad_f32": (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "spei_relu_bwd": (I, [P, P, P, L, P]),
    "spei_plane_ws_floats": (L, [I, I, I]),
    "spei_plane_stats": (I, [P, P, I, I, I, I, P, P, P, P, P, P, P]),
    "spei_plane_stats_batched": (I, [P, P, I, I, I, I, P, P, P, P, P, P, I, P]),
    "spei_resblock_apply_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "spei_resblock_apply_bwd_batched": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "spei_gate_train_saved_floats": (L, [I, I, I, I, I]),
    "spei_gate_train_ws_floats": (L, [I, I, I, I, I]),
    "spei_gate_train_nparams": (I, [I]),
    "spei_gate_maps_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, P, P]),
    "spei_gate_maps_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "spei_ln_bwd_blocks": (L, [L]),
    "spei_layernorm256_bwd": (I, [P, P, P, P, P, L, P]),
    "spei_gelu_fwd": (I, [P, P, L, P]),
    "spei_gelu_bwd": (I, [P, P, P, L, P]),
    "spei_window_attention_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    "spei_scale_rows": (I, [P, P, P, L, I, P]),
    "spei_corr_s_bwd_lr": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "spei_search_bwd_ref": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "spei_upsample_bicubic_bwd": (I, [P, P, I, I, I, I, P]),
    "spei_rowdot": (I, [P, P, P, L, I, P]),
    "spei_frame_post_ws_doubles": (L, [I, I, I]),
    "spei_frame_post": (I, [P, P, P, I, I, I, P, P, P]),
    "spei_det_gray": (I, [P, P, I, I, I, P]),
    "spei_det_ws_floats": (L, [I, I, I, I]),
    "spei_det_features": (I, [P, P, P, I, I, I, I, P]),
}


def header_symbols() -> list:
    """Entry points declared in include/speinet_hip.h (parsed from the header text)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spei_[a-z0-9_]+)\s*\(", text)) - {"spei_stream_t"})


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m speinet_amd.build` (hipcc --offload-arch=gfx950). "
                "speinet_amd has no CPU fallback.")
        # torch ships its own libamdhip64.so.7; it must be in the process BEFORE this library is loaded so
        # both resolve to ONE HIP runtime (same SONAME) and share device pointers and streams.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().spei_last_error().decode()}")
