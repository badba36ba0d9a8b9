"""ctypes binding of libullsam_hip.so (declared in include/ullsam_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ULLSAM_HIP_LIB") or os.path.join(_HERE, "lib", "libullsam_hip.so")  # env: A/B a side build (developer switch)

ABI_VERSION = 11  # == ULLSAM_ABI_VERSION in include/ullsam_hip.h (tests/test_host_cpu.py checks the three agree)

_lib = None

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_long, C.c_float

# name -> argtypes, exactly mirroring include/ullsam_hip.h
SIGNATURES = {
    "ullsam_gemm": [i32, vp, i64, vp, i64, vp, i64, i32, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, vp],
    "ullsam_gemm_qkv_rope": [i32, vp, i64, vp, i64, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, i64, vp],
    "ullsam_gemm_rmsnorm": [vp, i64, vp, f32, vp, i64, vp, i64, i32, vp, vp, i64, i32, i32, i32, i32, vp],
    "ullsam_decode_qkv_rope": [vp, vp, i64, vp, f32, vp, i64, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp],
    "ullsam_train_matmul": [vp, vp, vp, i32, i32, i32, i32] + [i64] * 9 + [i32, vp],
    "ullsam_train_matmul_bf16": [vp, vp, vp, i32, i32, i32, i32] + [i64] * 9 + [i32, vp],
    "ullsam_train_matmul_splitk": [vp, vp, vp, i32, i32, i32, i32] + [i64] * 9 + [i32, i32, vp, vp],
    "ullsam_train_matmul_heads": [vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i32, i64, i64, i64, i64, i32, i64, i64, i64, i64, i64, i64, i32, i32, i32, vp],
    "ullsam_train_colsum": [vp, vp, i64, i32, i64, vp, vp],
    "ullsam_train_ln_bwd": [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp, vp],
    "ullsam_train_act": [vp, vp, vp, i64, i32, vp],
    "ullsam_train_scale_shift": [vp, vp, vp, vp, vp, vp, vp, i64, vp, vp],
    "ullsam_train_attention": [vp] * 8 + [i32] * 7 + [vp] + [i64] * 12 + [f32, vp, vp, vp, vp, i32, vp],
    "ullsam_train_col2im3x3": [vp, vp, i32, i32, i32, i32, vp],
    "ullsam_train_attn_rows": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "ullsam_train_rmsnorm_bwd": [vp, vp, vp, vp, vp, i64, i32, f32, vp],
    "ullsam_train_rope": [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp],
    "ullsam_train_swiglu": [vp, vp, vp, vp, vp, vp, i64, vp],
    "ullsam_train_resize_bwd": [vp, vp, i64, i32, i32, i32, i32, vp],
    "ullsam_train_seg_loss": [vp, vp, vp, vp, i32, i64, f32, vp, vp],
    "ullsam_train_seg_loss_bwd": [vp, vp, vp, vp, vp, i32, i64, f32, vp],
    "ullsam_train_cross_entropy": [vp, i64, vp, vp, vp, vp, i64, i32, vp],
    "ullsam_train_cross_entropy_bwd": [vp, i64, vp, vp, vp, vp, vp, i64, i64, i32, vp],
    "ullsam_train_index_add_rows": [vp, vp, vp, i64, i32, i32, vp],
    "ullsam_norm": [vp, i32, i64, vp, i32, i64, vp, vp, i64, i32, f32, i32, i32, vp, vp, vp],
    "ullsam_norm_fanout": [vp, i64, i32, vp, vp, f32, vp, vp, vp, i32, vp, i64, vp],
    "ullsam_vit_attention": [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "ullsam_causal_attention": [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "ullsam_naive_attention": [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32] + [i64] * 12 + [f32, vp],
    "ullsam_fewkeys_attention": [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i64, vp],
    "ullsam_decode_attention": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp],
    "ullsam_tok2img_attention": [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, f32, vp, i32, vp],
    "ullsam_patch_im2col": [i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "ullsam_im2col3x3": [i32, vp, vp, i32, i32, i32, i32, vp],
    "ullsam_add_cast": [vp, i32, i64, vp, i64, vp, i32, i64, i32, vp],
    "ullsam_transpose_f32": [vp, vp, i32, i32, i32, vp],
    "ullsam_transpose_to_bf16": [i32, vp, vp, i32, i32, i32, vp],
    "ullsam_cast_transpose_bf16": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "ullsam_pixel_shuffle_ln": [i32, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "ullsam_pixel_unshuffle": [vp, vp, i32, i32, i32, i32, vp],
    "ullsam_scan_image_tokens": [vp, vp, vp, i32, i32, C.c_longlong, vp],
    "ullsam_embed_tokens": [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, vp],
    "ullsam_gather_rows": [vp, vp, vp, i32, i32, i32, i32, vp],
    "ullsam_rope_split": [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "ullsam_argmax": [vp, vp, i32, i64, i64, vp],
    "ullsam_small_linear": [vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp],
    "ullsam_i2t_block": [vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, i64, vp, vp, vp, i32, i32, i32, f32, vp],
    "ullsam_kv_proj": [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp],
    "ullsam_up2_hyper_masks": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "ullsam_dec_tok_attn": [vp] * 14 + [f32, vp, vp, i32, i32, i32, i32, vp],
    "ullsam_dec_tok_mlp": [vp] * 10 + [f32] + [vp] * 6 + [f32] + [vp] * 4 + [i32, i32, i32, vp],
    "ullsam_dec_heads": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "ullsam_concat_token_rows": [vp, i32, vp, i32, vp, i32, i32, vp],
    "ullsam_up1_ln_gelu": [vp, vp, vp, vp, vp, f32, vp, i64, vp],
    "ullsam_skinny_linear": [vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp],
    "ullsam_sparse_embed": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp],
    "ullsam_dense_pe": [vp, vp, i32, i32, i32, vp],
    "ullsam_mask_downscale": [vp, vp, i32, i32, i32, i32, i32, i32] + [vp] * 10 + [vp],
    "ullsam_hyper_masks": [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "ullsam_resize_bilinear": [vp, i64, i32, i32, i32, vp, vp, i32, i32, i32, f32, vp],
    "ullsam_mask_iou_counts": [vp, vp, vp, i32, i64, vp],
    "ullsam_stability_score": [vp, i64, i64, f32, f32, vp, vp, vp],
    "ullsam_mask_to_box": [vp, i64, i32, i32, vp, vp],
    "ullsam_rle_pack": [vp, i64, i32, i32, vp, vp, vp, vp],
    "ullsam_rle_emit": [vp, vp, i64, i32, i32, vp, vp, vp],
    "ullsam_amg_postprocess": [vp, vp, i64] + [i32] * 11 + [f32, f32, vp, vp, vp, vp, vp, vp],
    "ullsam_nms_mask": [vp, i32, f32, vp, vp],
    "ullsam_threshold_u8": [vp, vp, i64, f32, vp],
    "ullsam_rows_fp8": [vp, i32, i64, vp, i64, vp, vp, vp, i64, i32, f32, vp],
    "ullsam_gemm_fp8": [vp, i64, vp, vp, i64, vp, vp, i64, i32, vp, vp, i64, i32, i32, i32, i32, vp],
}
PLAIN = {"ullsam_last_error_string": ([], C.c_char_p), "ullsam_abi_version": ([], i32), "ullsam_device_count": ([], i32),
         "ullsam_set_gemm_variant": ([i32], i32), "ullsam_set_gemm_tuning": ([i32, i32], i32), "ullsam_set_attn_variant": ([i32], i32),
         "ullsam_set_norm_variant": ([i32], i32), "ullsam_train_set_matmul_mfma": ([i32], i32), "ullsam_train_set_matmul_vec": ([i32], i32), "ullsam_train_set_rows_reg": ([i32], i32), "ullsam_set_skinny_linear_mfma": ([i32], i32), "ullsam_set_attn_debug": ([vp], i32)}


class UllsamError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UllsamError(f"{LIB_PATH} not found -- run `python -m ullsam_amd.build` (there is no CPU fallback)")
    # torch first: it ships its own libamdhip64 / libhsa-runtime64, and the process must hold ONE HIP runtime -- the one whose device
    # pointers and streams this library is handed.  Loaded before torch, the library would bind /opt/rocm's runtime instead, and the
    # second runtime in the process finds no device ("no ROCm-capable device is detected" on the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    for name, (args, res) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    got = lib.ullsam_abi_version()
    if got != ABI_VERSION:
        raise UllsamError(f"{LIB_PATH} reports ABI version {got}, this binding is version {ABI_VERSION}: stale build -- run `python -m ullsam_amd.build --force`")
    _lib = lib
    return lib


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.ullsam_last_error_string()
        raise UllsamError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")
