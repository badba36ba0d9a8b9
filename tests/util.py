"""Shared helpers for the tests: golden loading, seeded inputs, shape tables."""
import ast
import os

import numpy as np

from oracle import ullsam_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def rand_image(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).random(shape, dtype=np.float32) * scale).astype(np.float32)


def decoder_inputs(seed=2):
    """Same draw order as oracle/gen_golden.py::case_decoder."""
    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    llm = rng.standard_normal((1, 256, 64, 64), dtype=np.float32) * 3.0 + 0.5
    mask_in = rng.standard_normal((3, 1, 256, 256), dtype=np.float32)
    return emb, llm, mask_in


VIT_TINY = dict(img_size=160, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, out_chans=64,
                window_size=7, global_attn_indexes=(1,))
VIT_SMALL = dict(img_size=1024, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, out_chans=256,
                 window_size=14, global_attn_indexes=(1,))
VIT_B = dict(img_size=1024, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, out_chans=256,
             window_size=14, global_attn_indexes=(2, 5, 8, 11))
VIT_H_D2 = dict(img_size=1024, patch_size=16, embed_dim=1280, depth=2, num_heads=16, mlp_ratio=4, out_chans=256,
                window_size=14, global_attn_indexes=(1,))  # ViT-H width (build_sam.py:14-21), one windowed + one global block
LLM_7B_L1 = dict(hidden=4096, layers=1, heads=32, kv_heads=8, inter=14336, vocab=92553, rope_theta=1000000.0, eps=1e-5)
LLM_TINY = dict(hidden=256, layers=2, heads=2, kv_heads=1, inter=512, vocab=92553, rope_theta=1000000.0, eps=1e-5)


def vit_run_cfg(c):
    return dict(depth=c["depth"], num_heads=c["num_heads"], global_attn_indexes=c["global_attn_indexes"],
                window_size=c["window_size"])


def vit_params(c, seed=0, prefix=""):
    return O.fill_state(O.vit_shapes(embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["num_heads"],
                                     global_attn_indexes=c["global_attn_indexes"], img_size=c["img_size"],
                                     patch_size=c["patch_size"], window_size=c["window_size"],
                                     out_chans=c["out_chans"], prefix=prefix), seed)


def llm_params(c, seed=0, prefix="language_model."):
    return O.fill_state(O.internlm2_shapes(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["inter"],
                                           c["vocab"], prefix=prefix), seed)


def ullsam_tiny_params(seed=0):
    P = {}
    P.update(vit_params(VIT_SMALL, seed, "vision_model."))
    P.update(O.fill_state(O.prompt_encoder_shapes(prefix="prompt_encoder."), seed))
    P.update(O.fill_state(O.mask_decoder_shapes(prefix="mask_decoder."), seed))
    P.update(llm_params(LLM_TINY, seed))
    P.update(O.fill_state(O.projector_shapes(LLM_TINY["hidden"]), seed))
    return P


def llm_7b_l1_inputs(seed=8):
    """Same draw as oracle/gen_golden.py::case_llm_7b_l1."""
    return np.random.default_rng(seed).standard_normal((2, 1081, 4096), dtype=np.float32) * np.float32(0.5)
