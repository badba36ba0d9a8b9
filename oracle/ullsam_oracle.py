"""CPU oracle for the uLLSAM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy (float32) restatement of the reference algorithm
  SAM ViT image encoder -> pixel-shuffle + mlp1 -> InternLM2 prefill -> mlp2 + inverse
  shuffle -> prompt encoder -> two-way mask decoder -> bilinear upsample / threshold / IoU
plus a greedy-decode loop with a KV cache.  Every function cites the reference file:line
it follows (paths relative to /root/reference).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s cpu_baseline leg may import this module; the shipped package
(`ullsam_amd/`) never does -- it fails loudly when the HIP library is missing.

Parity pinning: the reference ships no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, imported in the build
container by `oracle/gen_golden.py`; the captured vectors live in `tests/golden/` and
`tests/test_oracle_golden.py` checks this file against every one of them.

Parameters are passed as flat dicts {reference state_dict key: np.ndarray(float32)} so the
same dict can be loaded into the reference (gen_golden.py), into this oracle, and into the
HIP-backed modules (`load_state_dict`).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

try:  # scipy ships in the image; erf is only needed for exact GELU
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float32])

F32 = np.float32
Params = Dict[str, np.ndarray]

IMG_CONTEXT_TOKEN_ID = 92546  # modeling/modeling_internvl_sam.py:102
EOS_TOKEN_ID = 92542  # id('<|im_end|>'), modeling/modeling_internvl_sam.py:288,309


# --------------------------------------------------------------------------------------
# deterministic weight filler (shared by gen_golden.py, the tests and smoke())
# --------------------------------------------------------------------------------------
def fill_param(name: str, shape: Sequence[int], seed: int = 0) -> np.ndarray:
    """Deterministic float32 values for parameter `name`.

    Independent stream per name (seed, crc32(name)) so any subset can be regenerated
    anywhere without the reference.  Zero-initialised reference parameters (pos_embed
    image_encoder.py:68-70, rel_pos_h/w :221-222, llm_bias prompt_encoder.py:51) get
    non-zero values on purpose, otherwise those code paths would be untested.
    """
    rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    x = rng.standard_normal(shape, dtype=np.float32)
    leaf = name.split(".")[-1]
    if "llm_scale_factor" in name:
        return (0.1 + 0.02 * x).astype(F32)
    if "llm_bias" in name:
        return (0.05 + 0.02 * x).astype(F32)
    if "positional_encoding_gaussian_matrix" in name:
        return x
    if "rel_pos" in name:
        return (0.1 * x).astype(F32)
    if "pos_embed" in name:
        return (0.05 * x).astype(F32)
    is_norm = ("norm" in name) or name.endswith("neck.1.weight") or name.endswith("neck.3.weight") \
        or name.endswith("neck.1.bias") or name.endswith("neck.3.bias") \
        or name.startswith(("mlp1.0.", "mlp2.0.")) or ".mlp1.0." in name or ".mlp2.0." in name \
        or "output_upscaling.1." in name or "mask_downscaling.1." in name or "mask_downscaling.4." in name
    if is_norm:
        if leaf == "weight":
            return (1.0 + 0.1 * x).astype(F32)
        return (0.05 * x).astype(F32)
    if leaf == "bias" or len(shape) == 1:
        return (0.05 * x).astype(F32)
    if "tok_embeddings" in name or "iou_token" in name or "mask_tokens" in name \
            or "point_embeddings" in name or "not_a_point_embed" in name or "no_mask_embed" in name:
        return (0.5 * x).astype(F32)
    if "output_upscaling" in name and len(shape) == 4:  # ConvTranspose2d [Cin, Cout, 2, 2]
        return (x / math.sqrt(shape[0])).astype(F32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return (x / math.sqrt(max(fan_in, 1))).astype(F32)


def fill_rule(name: str, shape: Sequence[int]) -> Tuple[float, float]:
    """(mean, std) of the normal distribution fill_param draws parameter `name` from -- the same table, stated as numbers, for
    the agreement check with bench.py's random init (ullsam_amd/utils/synthetic.py::param_init_rule; tests/test_host_cpu.py)."""
    shape = tuple(int(s) for s in shape)
    if int(np.prod(shape)) > (1 << 22) and len(shape) > 1:
        shape = (max(1, (1 << 22) // int(np.prod(shape[1:]))),) + shape[1:]   # big matrices: fewer rows (no rule depends on dim 0 of a big one)
    if int(np.prod(shape)) < 64:
        shape = (64,) + shape[1:]   # one-element parameters (llm_scale_factor, llm_bias): a line needs more than one point
    v = fill_param(name, shape, 0).astype(np.float64)
    z = np.random.default_rng([0, zlib.crc32(name.encode())]).standard_normal(v.shape, dtype=np.float32).astype(np.float64)
    # fill_param is affine in its N(0, 1) draw: recover (mean, std) exactly by a least-squares line through (z, v)
    zz, vv = z.reshape(-1), v.reshape(-1)
    std = float(((zz - zz.mean()) * (vv - vv.mean())).sum() / ((zz - zz.mean()) ** 2).sum())
    mean = float(vv.mean() - std * zz.mean())
    return float(mean), float(std)


def fill_state(shapes: Dict[str, Sequence[int]], seed: int = 0) -> Params:
    return {k: fill_param(k, s, seed) for k, s in shapes.items()}


# --------------------------------------------------------------------------------------
# parameter shape tables (state_dict key layout)
# --------------------------------------------------------------------------------------
def vit_shapes(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11),
               img_size=1024, patch_size=16, window_size=14, out_chans=256, mlp_ratio=4,
               in_chans=3, prefix="") -> Dict[str, Tuple[int, ...]]:
    g = img_size // patch_size
    hd = embed_dim // num_heads
    s: Dict[str, Tuple[int, ...]] = {}
    p = prefix
    s[p + "pos_embed"] = (1, g, g, embed_dim)
    s[p + "patch_embed.proj.weight"] = (embed_dim, in_chans, patch_size, patch_size)
    s[p + "patch_embed.proj.bias"] = (embed_dim,)
    for i in range(depth):
        b = f"{p}blocks.{i}."
        ws = 0 if i in global_attn_indexes else window_size
        n = g if ws == 0 else ws
        s[b + "norm1.weight"] = (embed_dim,)
        s[b + "norm1.bias"] = (embed_dim,)
        s[b + "attn.rel_pos_h"] = (2 * n - 1, hd)
        s[b + "attn.rel_pos_w"] = (2 * n - 1, hd)
        s[b + "attn.qkv.weight"] = (3 * embed_dim, embed_dim)
        s[b + "attn.qkv.bias"] = (3 * embed_dim,)
        s[b + "attn.proj.weight"] = (embed_dim, embed_dim)
        s[b + "attn.proj.bias"] = (embed_dim,)
        s[b + "norm2.weight"] = (embed_dim,)
        s[b + "norm2.bias"] = (embed_dim,)
        s[b + "mlp.lin1.weight"] = (int(embed_dim * mlp_ratio), embed_dim)
        s[b + "mlp.lin1.bias"] = (int(embed_dim * mlp_ratio),)
        s[b + "mlp.lin2.weight"] = (embed_dim, int(embed_dim * mlp_ratio))
        s[b + "mlp.lin2.bias"] = (embed_dim,)
    s[p + "neck.0.weight"] = (out_chans, embed_dim, 1, 1)
    s[p + "neck.1.weight"] = (out_chans,)
    s[p + "neck.1.bias"] = (out_chans,)
    s[p + "neck.2.weight"] = (out_chans, out_chans, 3, 3)
    s[p + "neck.3.weight"] = (out_chans,)
    s[p + "neck.3.bias"] = (out_chans,)
    return s


def prompt_encoder_shapes(embed_dim=256, mask_in_chans=16, prefix="") -> Dict[str, Tuple[int, ...]]:
    p = prefix
    s: Dict[str, Tuple[int, ...]] = {
        p + "llm_scale_factor": (1,),
        p + "llm_bias": (1,),
        p + "pe_layer.positional_encoding_gaussian_matrix": (2, embed_dim // 2),
        p + "not_a_point_embed.weight": (1, embed_dim),
        p + "no_mask_embed.weight": (1, embed_dim),
        p + "mask_downscaling.0.weight": (mask_in_chans // 4, 1, 2, 2),
        p + "mask_downscaling.0.bias": (mask_in_chans // 4,),
        p + "mask_downscaling.1.weight": (mask_in_chans // 4,),
        p + "mask_downscaling.1.bias": (mask_in_chans // 4,),
        p + "mask_downscaling.3.weight": (mask_in_chans, mask_in_chans // 4, 2, 2),
        p + "mask_downscaling.3.bias": (mask_in_chans,),
        p + "mask_downscaling.4.weight": (mask_in_chans,),
        p + "mask_downscaling.4.bias": (mask_in_chans,),
        p + "mask_downscaling.6.weight": (embed_dim, mask_in_chans, 1, 1),
        p + "mask_downscaling.6.bias": (embed_dim,),
    }
    for i in range(4):
        s[p + f"point_embeddings.{i}.weight"] = (1, embed_dim)
    return s


def mask_decoder_shapes(dim=256, depth=2, mlp_dim=2048, num_mask_tokens=4, iou_hidden=256,
                        prefix="") -> Dict[str, Tuple[int, ...]]:
    p = prefix
    s: Dict[str, Tuple[int, ...]] = {p + "iou_token.weight": (1, dim), p + "mask_tokens.weight": (num_mask_tokens, dim)}

    def attn(b, internal):
        for n in ("q_proj", "k_proj", "v_proj"):
            s[b + n + ".weight"] = (internal, dim)
            s[b + n + ".bias"] = (internal,)
        s[b + "out_proj.weight"] = (dim, internal)
        s[b + "out_proj.bias"] = (dim,)

    for i in range(depth):
        b = f"{p}transformer.layers.{i}."
        attn(b + "self_attn.", dim)
        attn(b + "cross_attn_token_to_image.", dim // 2)
        attn(b + "cross_attn_image_to_token.", dim // 2)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            s[b + n + ".weight"] = (dim,)
            s[b + n + ".bias"] = (dim,)
        s[b + "mlp.lin1.weight"] = (mlp_dim, dim)
        s[b + "mlp.lin1.bias"] = (mlp_dim,)
        s[b + "mlp.lin2.weight"] = (dim, mlp_dim)
        s[b + "mlp.lin2.bias"] = (dim,)
    attn(p + "transformer.final_attn_token_to_image.", dim // 2)
    s[p + "transformer.norm_final_attn.weight"] = (dim,)
    s[p + "transformer.norm_final_attn.bias"] = (dim,)
    s[p + "output_upscaling.0.weight"] = (dim, dim // 4, 2, 2)
    s[p + "output_upscaling.0.bias"] = (dim // 4,)
    s[p + "output_upscaling.1.weight"] = (dim // 4,)
    s[p + "output_upscaling.1.bias"] = (dim // 4,)
    s[p + "output_upscaling.3.weight"] = (dim // 4, dim // 8, 2, 2)
    s[p + "output_upscaling.3.bias"] = (dim // 8,)
    for i in range(num_mask_tokens):
        b = f"{p}output_hypernetworks_mlps.{i}.layers."
        s[b + "0.weight"] = (dim, dim); s[b + "0.bias"] = (dim,)
        s[b + "1.weight"] = (dim, dim); s[b + "1.bias"] = (dim,)
        s[b + "2.weight"] = (dim // 8, dim); s[b + "2.bias"] = (dim // 8,)
    b = p + "iou_prediction_head.layers."
    s[b + "0.weight"] = (iou_hidden, dim); s[b + "0.bias"] = (iou_hidden,)
    s[b + "1.weight"] = (iou_hidden, iou_hidden); s[b + "1.bias"] = (iou_hidden,)
    s[b + "2.weight"] = (num_mask_tokens, iou_hidden); s[b + "2.bias"] = (num_mask_tokens,)
    return s


def internlm2_shapes(hidden=2048, layers=24, heads=16, kv_heads=8, inter=8192, vocab=92553,
                     prefix="", bias=False) -> Dict[str, Tuple[int, ...]]:
    p = prefix
    hd = hidden // heads
    s: Dict[str, Tuple[int, ...]] = {p + "model.tok_embeddings.weight": (vocab, hidden)}
    for i in range(layers):
        b = f"{p}model.layers.{i}."
        s[b + "attention.wqkv.weight"] = ((heads + 2 * kv_heads) * hd, hidden)
        s[b + "attention.wo.weight"] = (hidden, heads * hd)
        if bias:  # config.bias (modeling_internlm2.py:300-308)
            s[b + "attention.wqkv.bias"] = ((heads + 2 * kv_heads) * hd,)
            s[b + "attention.wo.bias"] = (hidden,)
        s[b + "feed_forward.w1.weight"] = (inter, hidden)
        s[b + "feed_forward.w3.weight"] = (inter, hidden)
        s[b + "feed_forward.w2.weight"] = (hidden, inter)
        s[b + "attention_norm.weight"] = (hidden,)
        s[b + "ffn_norm.weight"] = (hidden,)
    s[p + "model.norm.weight"] = (hidden,)
    s[p + "output.weight"] = (vocab, hidden)
    return s


def projector_shapes(llm_hidden: int, sam_hidden=256, prefix="") -> Dict[str, Tuple[int, ...]]:
    p = prefix
    c = sam_hidden * 4
    return {
        p + "mlp1.0.weight": (c,), p + "mlp1.0.bias": (c,),
        p + "mlp1.1.weight": (llm_hidden, c), p + "mlp1.1.bias": (llm_hidden,),
        p + "mlp1.3.weight": (llm_hidden, llm_hidden), p + "mlp1.3.bias": (llm_hidden,),
        p + "mlp2.0.weight": (llm_hidden,), p + "mlp2.0.bias": (llm_hidden,),
        p + "mlp2.1.weight": (c, llm_hidden), p + "mlp2.1.bias": (c,),
        p + "mlp2.3.weight": (c, c), p + "mlp2.3.bias": (c,),
    }


# --------------------------------------------------------------------------------------
# basic ops
# --------------------------------------------------------------------------------------
def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    y = np.matmul(x.astype(F32, copy=False), w.T.astype(F32, copy=False))
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def layer_norm(x: np.ndarray, w: Optional[np.ndarray], b: Optional[np.ndarray], eps: float) -> np.ndarray:
    """torch.nn.LayerNorm over the last dim (biased variance)."""
    u = x.mean(-1, keepdims=True, dtype=F32)
    d = x - u
    v = (d * d).mean(-1, keepdims=True, dtype=F32)
    y = d / np.sqrt(v + F32(eps))
    if w is not None:
        y = y * w
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def gelu(x: np.ndarray) -> np.ndarray:
    """nn.GELU() default = exact erf form (common.py:21-26 uses act_layer=nn.GELU)."""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(F32)


def softmax(x: np.ndarray, axis: int = -1) -> np.ndarray:
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(F32)


def silu(x: np.ndarray) -> np.ndarray:
    return (x / (1.0 + np.exp(-x))).astype(F32)


# --------------------------------------------------------------------------------------
# SAM ViT image encoder  (modeling/image_encoder.py)
# --------------------------------------------------------------------------------------
def patch_embed(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """PatchEmbed.forward image_encoder.py:391-395: Conv2d(k=s=patch) then NCHW->NHWC.
    Stride == kernel, so the conv is a GEMM over non-overlapping patches."""
    B, C, H, W = x.shape
    D, _, p, _ = w.shape
    gh, gw = H // p, W // p
    cols = x.reshape(B, C, gh, p, gw, p).transpose(0, 2, 4, 1, 3, 5).reshape(B, gh, gw, C * p * p)
    return linear(cols, w.reshape(D, -1), b)


def interp_linear_rows(t: np.ndarray, n_out: int) -> np.ndarray:
    """F.interpolate(t[None].permute(0, 2, 1), size=n_out, mode="linear") back as rows (image_encoder.py:308-316; align_corners=False):
    out[o] = (1 - l) t[i0] + l t[i1] with src = max((o + 0.5) L / n_out - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, L - 1), l = src - i0."""
    L = t.shape[0]
    src = np.maximum((np.arange(n_out, dtype=F32) + F32(0.5)) * F32(L / n_out) - F32(0.5), F32(0.0)).astype(F32)
    i0 = np.minimum(src.astype(np.int64), L - 1)
    i1 = np.minimum(i0 + 1, L - 1)
    lam = (src - i0.astype(F32)).astype(F32)[:, None]
    return (t[i0] * (F32(1.0) - lam) + t[i1] * lam).astype(F32)


def get_rel_pos(q_size: int, k_size: int, rel_pos: np.ndarray) -> np.ndarray:
    """image_encoder.py:292-322, including the interpolation of a table whose length is not 2 max(q, k) - 1 (:306-318: a checkpoint whose
    tables were trained at another resolution; at the sizes the reference builds, 1024^2 input, the lengths agree and the branch is not taken)."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        rel_pos = interp_linear_rows(rel_pos, max_rel_dist)
    q_coords = np.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = np.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[rel.astype(np.int64)]


def vit_attention(x: np.ndarray, P: Params, pre: str, num_heads: int) -> np.ndarray:
    """Attention.forward image_encoder.py:224-240 with add_decomposed_rel_pos :325-361.
    x: [B', H, W, D] (B' = images or images*windows)."""
    Bp, H, W, D = x.shape
    hd = D // num_heads
    scale = F32(hd ** -0.5)
    qkv = linear(x.reshape(Bp, H * W, D), P[pre + "qkv.weight"], P[pre + "qkv.bias"])
    qkv = qkv.reshape(Bp, H * W, 3, num_heads, hd).transpose(2, 0, 3, 1, 4)  # 3,B',h,N,hd
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = np.matmul(q * scale, k.transpose(0, 1, 3, 2))  # B',h,N,N
    Rh = get_rel_pos(H, H, P[pre + "rel_pos_h"])  # [H, H, hd]
    Rw = get_rel_pos(W, W, P[pre + "rel_pos_w"])
    r_q = q.reshape(Bp, num_heads, H, W, hd)  # NOTE: unscaled q (:234 passes q, not q*scale)
    rel_h = np.einsum("bnhwc,hkc->bnhwk", r_q, Rh).astype(F32)
    rel_w = np.einsum("bnhwc,wkc->bnhwk", r_q, Rw).astype(F32)
    attn = attn.reshape(Bp, num_heads, H, W, H, W) + rel_h[..., :, None] + rel_w[..., None, :]
    attn = softmax(attn.reshape(Bp, num_heads, H * W, H * W), -1)
    o = np.matmul(attn, v)  # B',h,N,hd
    o = o.transpose(0, 2, 1, 3).reshape(Bp, H, W, D)
    return linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"])


def window_partition(x: np.ndarray, ws: int):
    """image_encoder.py:243-264 (zero pad bottom/right AFTER norm1 -> pad tokens are live keys)."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = np.pad(x, ((0, 0), (0, ph), (0, pw), (0, 0)))
    Hp, Wp = H + ph, W + pw
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).transpose(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(w: np.ndarray, ws: int, pad_hw, hw) -> np.ndarray:
    """image_encoder.py:267-289."""
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).transpose(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def vit_block(x: np.ndarray, P: Params, pre: str, num_heads: int, window_size: int, ln_eps: float) -> np.ndarray:
    """Block.forward image_encoder.py:166-182; MLPBlock common.py:13-26."""
    shortcut = x
    h = layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], ln_eps)
    if window_size > 0:
        H, W = h.shape[1], h.shape[2]
        h, pad_hw = window_partition(h, window_size)
    h = vit_attention(h, P, pre + "attn.", num_heads)
    if window_size > 0:
        h = window_unpartition(h, window_size, pad_hw, (H, W))
    x = shortcut + h
    m = layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"], ln_eps)
    m = gelu(linear(m, P[pre + "mlp.lin1.weight"], P[pre + "mlp.lin1.bias"]))
    m = linear(m, P[pre + "mlp.lin2.weight"], P[pre + "mlp.lin2.bias"])
    return (x + m).astype(F32)


def conv3x3_nhwc(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """Conv2d(k=3, pad=1, bias=False) on NHWC input, w: [Cout, Cin, 3, 3] (image_encoder.py:96-102)."""
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    cols = np.concatenate([xp[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)], axis=-1)
    wk = w.transpose(0, 2, 3, 1).reshape(w.shape[0], -1)  # [Cout, (ky,kx,cin)]
    return linear(cols, wk)


def vit_encoder(x: np.ndarray, P: Params, *, depth: int, num_heads: int, global_attn_indexes: Sequence[int],
                window_size: int = 14, ln_eps: float = 1e-6, prefix: str = "", nhwc: bool = False) -> np.ndarray:
    """ImageEncoderViT.forward image_encoder.py:106-116 (+ neck :88-104, LayerNorm2d common.py:31-43,
    LN eps 1e-6 from build_sam.py:72).  x: [B,3,S,S] -> [B,out,S/16,S/16] (or NHWC if nhwc)."""
    p = prefix
    h = patch_embed(x.astype(F32), P[p + "patch_embed.proj.weight"], P[p + "patch_embed.proj.bias"])
    h = h + P[p + "pos_embed"]
    for i in range(depth):
        ws = 0 if i in global_attn_indexes else window_size
        h = vit_block(h, P, f"{p}blocks.{i}.", num_heads, ws, ln_eps)
    w0 = P[p + "neck.0.weight"]
    h = linear(h, w0.reshape(w0.shape[0], -1))
    h = layer_norm(h, P[p + "neck.1.weight"], P[p + "neck.1.bias"], 1e-6)  # LayerNorm2d == LN over C in NHWC
    h = conv3x3_nhwc(h, P[p + "neck.2.weight"])
    h = layer_norm(h, P[p + "neck.3.weight"], P[p + "neck.3.bias"], 1e-6)
    return h if nhwc else np.ascontiguousarray(h.transpose(0, 3, 1, 2))


# --------------------------------------------------------------------------------------
# Prompt encoder (modeling/prompt_encoder.py)
# --------------------------------------------------------------------------------------
def _pe_encoding(coords01: np.ndarray, G: np.ndarray) -> np.ndarray:
    """PositionEmbeddingRandom._pe_encoding prompt_encoder.py:220-228."""
    c = (2.0 * coords01.astype(F32) - 1.0).astype(F32)
    c = np.matmul(c, G.astype(F32))
    c = (F32(2.0 * np.pi) * c).astype(F32)
    return np.concatenate([np.sin(c), np.cos(c)], axis=-1).astype(F32)


def dense_pe(P: Params, size: Tuple[int, int] = (64, 64), prefix: str = "") -> np.ndarray:
    """get_dense_pe / PositionEmbeddingRandom.forward prompt_encoder.py:65-74,230-241 -> [1,C,H,W]."""
    h, w = size
    y = ((np.arange(h, dtype=F32) + 0.5) / h)[:, None].repeat(w, 1)
    x = ((np.arange(w, dtype=F32) + 0.5) / w)[None, :].repeat(h, 0)
    pe = _pe_encoding(np.stack([x, y], -1), P[prefix + "pe_layer.positional_encoding_gaussian_matrix"])
    return pe.transpose(2, 0, 1)[None]


def prompt_encoder(P: Params, points: Optional[Tuple[np.ndarray, np.ndarray]], boxes: Optional[np.ndarray],
                   masks: Optional[np.ndarray], llm_hidden_states: Optional[np.ndarray] = None,
                   image_embedding_size=(64, 64), input_image_size=(1024, 1024), prefix: str = ""):
    """PromptEncoder.forward prompt_encoder.py:153-203 (_embed_points :76-94, _embed_boxes :96-103,
    _normalize_llm_hidden_states :131-151, _embed_masks :105-108)."""
    p = prefix
    G = P[p + "pe_layer.positional_encoding_gaussian_matrix"]
    C = G.shape[1] * 2
    if points is not None:
        bs = points[0].shape[0]
    elif boxes is not None:
        bs = boxes.shape[0]
    elif masks is not None:
        bs = masks.shape[0]
    else:
        bs = 1
    sparse = np.zeros((bs, 0, C), F32)

    def with_coords(c):
        c = c.astype(F32).copy()
        c[..., 0] = c[..., 0] / input_image_size[1]
        c[..., 1] = c[..., 1] / input_image_size[0]
        return _pe_encoding(c, G)

    if points is not None:
        coords, labels = points
        coords = coords.astype(F32) + 0.5
        labels = labels.astype(np.int64)
        if boxes is None:  # pad=(boxes is None) :181
            coords = np.concatenate([coords, np.zeros((bs, 1, 2), F32)], 1)
            labels = np.concatenate([labels, -np.ones((bs, 1), np.int64)], 1)
        pe = with_coords(coords)
        pe[labels == -1] = 0.0
        pe[labels == -1] += P[p + "not_a_point_embed.weight"][0]
        pe[labels == 0] += P[p + "point_embeddings.0.weight"][0]
        pe[labels == 1] += P[p + "point_embeddings.1.weight"][0]
        sparse = np.concatenate([sparse, pe], 1)
    if boxes is not None:
        c = (boxes.astype(F32) + 0.5).reshape(-1, 2, 2)
        ce = with_coords(c)
        ce[:, 0, :] += P[p + "point_embeddings.2.weight"][0]
        ce[:, 1, :] += P[p + "point_embeddings.3.weight"][0]
        sparse = np.concatenate([sparse, ce], 1)

    if masks is not None:
        dense = _mask_downscaling(masks.astype(F32), P, p)
    elif llm_hidden_states is not None:
        x = llm_hidden_states.astype(F32)  # [B,C,H,W]
        n = layer_norm(x.transpose(0, 2, 3, 1), None, None, 1e-5)  # F.layer_norm default eps, no affine
        n = n.transpose(0, 3, 1, 2) * P[p + "llm_scale_factor"] + P[p + "llm_bias"]
        dense = n.reshape(bs, -1, image_embedding_size[0], image_embedding_size[1]).astype(F32)
    else:
        dense = np.broadcast_to(P[p + "no_mask_embed.weight"].reshape(1, -1, 1, 1),
                                (bs, C, image_embedding_size[0], image_embedding_size[1])).astype(F32)
    return sparse.astype(F32), np.ascontiguousarray(dense)


def _conv_k2s2(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Conv2d(kernel=2, stride=2) NCHW."""
    B, C, H, W = x.shape
    cols = x.reshape(B, C, H // 2, 2, W // 2, 2).transpose(0, 2, 4, 1, 3, 5).reshape(B, H // 2, W // 2, C * 4)
    y = linear(cols, w.reshape(w.shape[0], -1), b)
    return y.transpose(0, 3, 1, 2)


def _ln2d(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """LayerNorm2d common.py:38-43 on NCHW."""
    return layer_norm(x.transpose(0, 2, 3, 1), w, b, eps).transpose(0, 3, 1, 2)


def _mask_downscaling(m: np.ndarray, P: Params, p: str) -> np.ndarray:
    """prompt_encoder.py:54-62."""
    x = _conv_k2s2(m, P[p + "mask_downscaling.0.weight"], P[p + "mask_downscaling.0.bias"])
    x = gelu(_ln2d(x, P[p + "mask_downscaling.1.weight"], P[p + "mask_downscaling.1.bias"]))
    x = _conv_k2s2(x, P[p + "mask_downscaling.3.weight"], P[p + "mask_downscaling.3.bias"])
    x = gelu(_ln2d(x, P[p + "mask_downscaling.4.weight"], P[p + "mask_downscaling.4.bias"]))
    w = P[p + "mask_downscaling.6.weight"]
    x = linear(x.transpose(0, 2, 3, 1), w.reshape(w.shape[0], -1), P[p + "mask_downscaling.6.bias"])
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


# --------------------------------------------------------------------------------------
# Two-way transformer + mask decoder (modeling/transformer.py, modeling/mask_decoder.py)
# --------------------------------------------------------------------------------------
def _dec_attention(P: Params, pre: str, q, k, v, num_heads: int) -> np.ndarray:
    """transformer.Attention.forward transformer.py:220-242 (scale applied AFTER QK^T :233-235)."""
    q = linear(q, P[pre + "q_proj.weight"], P[pre + "q_proj.bias"])
    k = linear(k, P[pre + "k_proj.weight"], P[pre + "k_proj.bias"])
    v = linear(v, P[pre + "v_proj.weight"], P[pre + "v_proj.bias"])

    def sep(x):
        b, n, c = x.shape
        return x.reshape(b, n, num_heads, c // num_heads).transpose(0, 2, 1, 3)

    q, k, v = sep(q), sep(k), sep(v)
    c = q.shape[-1]
    a = np.matmul(q, k.transpose(0, 1, 3, 2)) / F32(math.sqrt(c))
    a = softmax(a, -1)
    o = np.matmul(a, v).transpose(0, 2, 1, 3)
    o = o.reshape(o.shape[0], o.shape[1], -1)
    return linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


def two_way_transformer(P: Params, pre: str, image_embedding: np.ndarray, image_pe: np.ndarray,
                        point_embedding: np.ndarray, depth: int = 2, num_heads: int = 8):
    """TwoWayTransformer.forward transformer.py:62-108; TwoWayAttentionBlock.forward :153-184."""
    bs, c, h, w = image_embedding.shape
    keys = image_embedding.reshape(bs, c, h * w).transpose(0, 2, 1)
    key_pe = image_pe.reshape(bs, c, h * w).transpose(0, 2, 1)
    queries = point_embedding
    query_pe = point_embedding
    ln = lambda x, n: layer_norm(x, P[n + ".weight"], P[n + ".bias"], 1e-5)
    for i in range(depth):
        b = f"{pre}layers.{i}."
        if i == 0:  # skip_first_layer_pe: no PE and NO residual (:157-158)
            queries = _dec_attention(P, b + "self_attn.", queries, queries, queries, num_heads)
        else:
            q = queries + query_pe
            queries = queries + _dec_attention(P, b + "self_attn.", q, q, queries, num_heads)
        queries = ln(queries, b + "norm1")
        q = queries + query_pe
        k = keys + key_pe
        queries = queries + _dec_attention(P, b + "cross_attn_token_to_image.", q, k, keys, num_heads)
        queries = ln(queries, b + "norm2")
        m = np.maximum(linear(queries, P[b + "mlp.lin1.weight"], P[b + "mlp.lin1.bias"]), 0)
        queries = queries + linear(m, P[b + "mlp.lin2.weight"], P[b + "mlp.lin2.bias"])
        queries = ln(queries, b + "norm3")
        q = queries + query_pe
        k = keys + key_pe
        keys = keys + _dec_attention(P, b + "cross_attn_image_to_token.", k, q, queries, num_heads)
        keys = ln(keys, b + "norm4")
    q = queries + query_pe
    k = keys + key_pe
    queries = queries + _dec_attention(P, pre + "final_attn_token_to_image.", q, k, keys, num_heads)
    queries = ln(queries, pre + "norm_final_attn")
    return queries.astype(F32), keys.astype(F32)


def _conv_transpose_k2s2(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """ConvTranspose2d(kernel=2, stride=2): w [Cin, Cout, 2, 2]; stride == kernel so every output
    pixel receives exactly one tap (mask_decoder.py:53-59)."""
    B, C, H, W = x.shape
    Co = w.shape[1]
    y = np.einsum("bchw,cokl->bohkwl", x, w).astype(F32)  # [B,Co,H,2,W,2]
    return y.reshape(B, Co, 2 * H, 2 * W) + b[None, :, None, None]


def _mlp_relu(P: Params, pre: str, x: np.ndarray, n: int = 3) -> np.ndarray:
    """mask_decoder.MLP.forward mask_decoder.py:171-176."""
    for i in range(n):
        x = linear(x, P[f"{pre}layers.{i}.weight"], P[f"{pre}layers.{i}.bias"])
        if i < n - 1:
            x = np.maximum(x, 0)
    return x


def mask_decoder(P: Params, image_embeddings: np.ndarray, image_pe: np.ndarray, sparse: np.ndarray,
                 dense: np.ndarray, multimask_output: bool, prefix: str = "", num_mask_tokens: int = 4):
    """MaskDecoder.forward / predict_masks mask_decoder.py:71-149."""
    p = prefix
    out_tokens = np.concatenate([P[p + "iou_token.weight"], P[p + "mask_tokens.weight"]], 0)
    nb = sparse.shape[0]
    tokens = np.concatenate([np.broadcast_to(out_tokens[None], (nb,) + out_tokens.shape), sparse], 1).astype(F32)
    src = np.repeat(image_embeddings, nb, axis=0) + dense
    pos = np.repeat(image_pe, nb, axis=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(P, p + "transformer.", src.astype(F32), pos.astype(F32), tokens)
    iou_tok = hs[:, 0, :]
    mask_toks = hs[:, 1:1 + num_mask_tokens, :]
    src = src.transpose(0, 2, 1).reshape(b, c, h, w)
    u = _conv_transpose_k2s2(src, P[p + "output_upscaling.0.weight"], P[p + "output_upscaling.0.bias"])
    u = gelu(_ln2d(u, P[p + "output_upscaling.1.weight"], P[p + "output_upscaling.1.bias"]))
    u = gelu(_conv_transpose_k2s2(u, P[p + "output_upscaling.3.weight"], P[p + "output_upscaling.3.bias"]))
    hyper = np.stack([_mlp_relu(P, f"{p}output_hypernetworks_mlps.{i}.", mask_toks[:, i, :])
                      for i in range(num_mask_tokens)], 1)
    bb, cc, hh, ww = u.shape
    masks = np.matmul(hyper, u.reshape(bb, cc, hh * ww)).reshape(bb, -1, hh, ww)
    iou = _mlp_relu(P, p + "iou_prediction_head.", iou_tok)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl].astype(F32), iou[:, sl].astype(F32)


def bilinear_resize(x: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """F.interpolate(mode='bilinear', align_corners=False) on [...,H,W] (app.py:635-640, sam.py:154-161)."""
    H, W = x.shape[-2:]
    oh, ow = out_hw

    def axis(n_in, n_out):
        s = n_in / n_out
        src = (np.arange(n_out, dtype=np.float64) + 0.5) * s - 0.5
        src = np.maximum(src, 0.0)
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0).astype(F32)
        return i0, i1, l1

    y0, y1, ly = axis(H, oh)
    x0, x1, lx = axis(W, ow)
    top = x[..., y0, :] * (1 - ly)[:, None] + x[..., y1, :] * ly[:, None]
    out = top[..., :, x0] * (1 - lx) + top[..., :, x1] * lx
    return out.astype(F32)


def calc_iou(pred: np.ndarray, gt: np.ndarray) -> float:
    """CalcIoU train_joint_v2.py:666-696 on boolean masks: (inter + 1e-7)/(union + 1e-7)."""
    p = pred.astype(bool).reshape(-1)
    g = gt.astype(bool).reshape(-1)
    inter = float(np.logical_and(p, g).sum())
    union = float(np.logical_or(p, g).sum())
    return (inter + 1e-7) / (union + 1e-7)


def sam_forward_one(P: Params, image: np.ndarray, point_coords, point_labels, multimask_output: bool,
                    vit_cfg: dict, pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)):
    """Sam.forward sam.py:53-131 for one image record with point prompts (image already HxW<=1024 in 0..255).
    Parameter prefixes: image_encoder. / prompt_encoder. / mask_decoder."""
    mean = np.asarray(pixel_mean, F32).reshape(3, 1, 1)
    std = np.asarray(pixel_std, F32).reshape(3, 1, 1)
    x = (image.astype(F32) - mean) / std
    h, w = x.shape[-2:]
    x = np.pad(x, ((0, 0), (0, 1024 - h), (0, 1024 - w)))
    emb = vit_encoder(x[None], P, prefix="image_encoder.", **vit_cfg)
    sp, de = prompt_encoder(P, (point_coords, point_labels), None, None, prefix="prompt_encoder.")
    low, iou = mask_decoder(P, emb, dense_pe(P, prefix="prompt_encoder."), sp, de, multimask_output,
                            prefix="mask_decoder.")
    up = bilinear_resize(low, (1024, 1024))[..., :h, :w]
    return {"masks": up > 0.0, "iou_predictions": iou, "low_res_logits": low}


# --------------------------------------------------------------------------------------
# InternLM2 (modeling/modeling_internlm2.py)
# --------------------------------------------------------------------------------------
def rms_norm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """InternLM2RMSNorm.forward modeling_internlm2.py:138-143 (fp32 throughout in the fp32 oracle)."""
    v = (x.astype(F32) ** 2).mean(-1, keepdims=True, dtype=F32)
    return (w * (x * (1.0 / np.sqrt(v + F32(eps))))).astype(F32)


def rope_tables(head_dim: int, n_pos: int, base: float, scaling: Optional[dict] = None, max_pos: int = 32768,
                seq_len: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """InternLM2RotaryEmbedding modeling_internlm2.py:147-180 (fp32 cache); `scaling` = config.rope_scaling:
    {"type": "linear"} divides the positions by the factor (:184-200); {"type": "dynamic"} rescales the base once the sequence
    length `seq_len` of the call that (re)builds the cache exceeds max_position_embeddings (:204-229)."""
    scale_t = F32(1.0)
    if scaling is not None:
        f = float(scaling["factor"])
        if scaling["type"] == "linear":
            scale_t = F32(f)
        else:
            sl = n_pos if seq_len is None else seq_len
            if sl > max_pos:
                base = float(base) * ((f * sl / max_pos) - (f - 1)) ** (head_dim / (head_dim - 2))
    inv = (1.0 / (F32(base) ** (np.arange(0, head_dim, 2, dtype=F32) / F32(head_dim)))).astype(F32)
    t = np.arange(n_pos, dtype=F32) / scale_t
    fr = np.einsum("i,j->ij", t, inv).astype(F32)
    emb = np.concatenate([fr, fr], -1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], -1)


def internlm2_attention(P: Params, pre: str, x: np.ndarray, mask4d: Optional[np.ndarray], pos_ids: np.ndarray,
                        past: Optional[Tuple[np.ndarray, np.ndarray]], cfg: dict):
    """InternLM2Attention.forward modeling_internlm2.py:341-426 (eager path); repeat_kv :268-277."""
    B, S, _ = x.shape
    H, KV, hd = cfg["heads"], cfg["kv_heads"], cfg["hidden"] // cfg["heads"]
    g = H // KV
    qkv = linear(x, P[pre + "wqkv.weight"], P.get(pre + "wqkv.bias")).reshape(B, S, KV, g + 2, hd)  # 'b q (h gs d)' :361-366
    q = qkv[..., :g, :].reshape(B, S, H, hd).transpose(0, 2, 1, 3)
    k = qkv[..., -2, :].transpose(0, 2, 1, 3)
    v = qkv[..., -1, :].transpose(0, 2, 1, 3)
    kv_len = S + (past[0].shape[2] if past is not None else 0)
    cos, sin = rope_tables(hd, max(kv_len, int(pos_ids.max()) + 1), cfg["rope_theta"], cfg.get("rope_scaling"),
                           cfg.get("max_pos", 32768), seq_len=kv_len)
    c = cos[pos_ids][:, None]
    s = sin[pos_ids][:, None]
    q = q * c + _rotate_half(q) * s
    k = k * c + _rotate_half(k) * s
    if past is not None:
        k = np.concatenate([past[0], k], 2)
        v = np.concatenate([past[1], v], 2)
    present = (k, v)
    kr = np.repeat(k, g, axis=1)
    vr = np.repeat(v, g, axis=1)
    a = np.matmul(q, kr.transpose(0, 1, 3, 2)) / F32(math.sqrt(hd))
    if mask4d is not None:
        a = a + mask4d
    a = softmax(a.astype(F32), -1)
    o = np.matmul(a, vr).transpose(0, 2, 1, 3).reshape(B, S, H * hd)
    return linear(o, P[pre + "wo.weight"], P.get(pre + "wo.bias")), present


def decoder_mask(attention_mask: np.ndarray, q_len: int, past_len: int) -> Optional[np.ndarray]:
    """_prepare_decoder_attention_mask modeling_internlm2.py:830-851 with _make_causal_mask :96-110
    and _expand_mask :114-125 (additive finfo(fp32).min masks, summed)."""
    mn = np.finfo(np.float32).min
    B, src = attention_mask.shape
    comb = None
    if q_len > 1:
        cm = np.full((q_len, q_len), mn, F32)
        cm[np.tril_indices(q_len)] = 0
        if past_len > 0:
            cm = np.concatenate([np.zeros((q_len, past_len), F32), cm], -1)
        comb = np.broadcast_to(cm[None, None], (B, 1, q_len, q_len + past_len))
    inv = 1.0 - attention_mask[:, None, None, :].astype(F32)
    exp = np.where(inv.astype(bool), F32(mn), inv).astype(F32)
    exp = np.broadcast_to(exp, (B, 1, q_len, src))
    with np.errstate(over="ignore"):
        return (exp if comb is None else exp + comb).astype(F32)


def internlm2_model(P: Params, cfg: dict, inputs_embeds: np.ndarray, attention_mask: Optional[np.ndarray] = None,
                    position_ids: Optional[np.ndarray] = None, past=None, use_cache: bool = False,
                    prefix: str = ""):
    """InternLM2Model.forward modeling_internlm2.py:854-984 (+ DecoderLayer :621-681, MLP :261-264).
    Returns (post-final-norm hidden [B,S,D], new past or None)."""
    p = prefix
    B, S, _ = inputs_embeds.shape
    past_len = past[0][0].shape[2] if past is not None else 0
    if position_ids is None:
        position_ids = np.arange(past_len, S + past_len, dtype=np.int64)[None].repeat(B, 0)
    if attention_mask is None:
        attention_mask = np.ones((B, S + past_len), np.int64)
    m4 = decoder_mask(attention_mask, S, past_len)
    h = inputs_embeds.astype(F32)
    new_past = []
    for i in range(cfg["layers"]):
        b = f"{p}model.layers.{i}."
        r = h
        a, pres = internlm2_attention(P, b + "attention.", rms_norm(h, P[b + "attention_norm.weight"], cfg["eps"]),
                                      m4, position_ids, past[i] if past is not None else None, cfg)
        h = r + a
        r = h
        n = rms_norm(h, P[b + "ffn_norm.weight"], cfg["eps"])
        f = linear(silu(linear(n, P[b + "feed_forward.w1.weight"])) * linear(n, P[b + "feed_forward.w3.weight"]),
                   P[b + "feed_forward.w2.weight"])
        h = (r + f).astype(F32)
        if use_cache:
            new_past.append(pres)
    h = rms_norm(h, P[p + "model.norm.weight"], cfg["eps"])
    return h, (new_past if use_cache else None)


def lm_head(P: Params, hidden: np.ndarray, prefix: str = "") -> np.ndarray:
    """InternLM2ForCausalLM.forward modeling_internlm2.py:1080-1082."""
    return linear(hidden, P[prefix + "output.weight"])


# --------------------------------------------------------------------------------------
# InternVLSAMModel composite (modeling/modeling_internvl_sam.py)
# --------------------------------------------------------------------------------------
def pixel_shuffle_v2(x: np.ndarray, scale: float = 0.5) -> np.ndarray:
    """pixel_shuffle modeling_internvl_sam.py:226-240 with ps_version='v2' (train_joint_v2.py:1424-1431)."""
    n, h, w, c = x.shape
    x = x.reshape(n, h, int(w * scale), int(c / scale)).transpose(0, 2, 1, 3)
    x = x.reshape(n, int(w * scale), int(h * scale), int(c / (scale * scale))).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(x)


def extract_feature(P: Params, vit_features_nchw: np.ndarray) -> np.ndarray:
    """extract_feature :242-251 + mlp1 :88-93 (LN eps 1e-5 default, exact GELU)."""
    f = pixel_shuffle_v2(vit_features_nchw.transpose(0, 2, 3, 1))
    f = f.reshape(f.shape[0], -1, f.shape[-1])
    f = layer_norm(f, P["mlp1.0.weight"], P["mlp1.0.bias"], 1e-5)
    f = gelu(linear(f, P["mlp1.1.weight"], P["mlp1.1.bias"]))
    return linear(f, P["mlp1.3.weight"], P["mlp1.3.bias"])


def text_aware_dense_feature(P: Params, feats: np.ndarray, ratio: float = 0.5) -> np.ndarray:
    """text_aware_dense_feature :253-270 + mlp2 :95-100 -> [B,256,64,64]."""
    f = layer_norm(feats, P["mlp2.0.weight"], P["mlp2.0.bias"], 1e-5)
    f = gelu(linear(f, P["mlp2.1.weight"], P["mlp2.1.bias"]))
    f = linear(f, P["mlp2.3.weight"], P["mlp2.3.bias"])
    s = int(math.sqrt(f.shape[1]))
    f = f.reshape(f.shape[0], s, s, f.shape[2]).transpose(0, 2, 1, 3)  # ps_version != 'v1'
    n, h, w, c = f.shape
    f = f.reshape(n, h, int(w / ratio), int(c * ratio)).transpose(0, 2, 1, 3)
    f = f.reshape(n, int(w / ratio), int(h / ratio), int(c * ratio * ratio))
    return np.ascontiguousarray(f.transpose(0, 3, 1, 2))


def build_inputs_embeds(P: Params, input_ids: np.ndarray, vit_embeds: np.ndarray, prefix="language_model.") -> np.ndarray:
    """Token scatter, forward :124-158 (rows where ids == 92546 <- vit_embeds rows in order).
    Applied per sample; at B=1 identical to the reference's flattened form."""
    emb = P[prefix + "model.tok_embeddings.weight"][input_ids].astype(F32).copy()
    for b in range(input_ids.shape[0]):
        sel = np.nonzero(input_ids[b] == IMG_CONTEXT_TOKEN_ID)[0]
        fv = vit_embeds[b].reshape(-1, emb.shape[-1])
        reps = (len(sel) + fv.shape[0] - 1) // max(fv.shape[0], 1)
        if reps > 1:
            fv = np.tile(fv, (reps, 1))
        emb[b, sel] = fv[:len(sel)]
    return emb


def ullsam_forward(P: Params, pixel_values: np.ndarray, input_ids: np.ndarray, attention_mask: Optional[np.ndarray],
                   vit_cfg: dict, llm_cfg: dict, want_logits: bool = False):
    """InternVLSAMModel.forward modeling_internvl_sam.py:106-224 with output_hidden_states=True.
    B>1 is defined as the reference run per sample at B=1 (the reference itself fails at B>1,
    SURVEY.md section 0).  Returns dict(image_embeddings, vit_embeds, last_hidden, dense_feature[, logits])."""
    img = vit_encoder(pixel_values, P, prefix="vision_model.", **vit_cfg)
    vit_embeds = extract_feature(P, img)
    emb = build_inputs_embeds(P, input_ids, vit_embeds)
    hidden, _ = internlm2_model(P, llm_cfg, emb, attention_mask, prefix="language_model.")
    feats = []
    for b in range(input_ids.shape[0]):
        idx = np.nonzero(input_ids[b] == IMG_CONTEXT_TOKEN_ID)[0]
        if len(idx) == 0:
            raise ValueError("Can not find vision token!")
        feats.append(hidden[b, idx.min():idx.max() + 1])
    dense = text_aware_dense_feature(P, np.stack(feats, 0))
    out = {"image_embeddings": img, "vit_embeds": vit_embeds, "last_hidden": hidden, "dense_feature": dense}
    if want_logits:
        out["logits"] = lm_head(P, hidden, "language_model.")
    return out


def ullsam_mask_path(P: Params, pixel_values: np.ndarray, input_ids: np.ndarray, points: np.ndarray,
                     labels: np.ndarray, vit_cfg: dict, llm_cfg: dict, use_llm_dense: bool = True):
    """The metric path, app.py:580-645: forward -> prompt encoder -> mask decoder -> x4 bilinear -> threshold.
    One prompt set per image (points [B,Np,2], labels [B,Np])."""
    f = ullsam_forward(P, pixel_values, input_ids, None, vit_cfg, llm_cfg)
    pe = dense_pe(P, prefix="prompt_encoder.")
    lows, ious = [], []
    for b in range(pixel_values.shape[0]):
        sp, de = prompt_encoder(P, (points[b:b + 1], labels[b:b + 1]), None, None,
                                f["dense_feature"][b:b + 1] if use_llm_dense else None, prefix="prompt_encoder.")
        low, iou = mask_decoder(P, f["image_embeddings"][b:b + 1], pe, sp, de, False, prefix="mask_decoder.")
        lows.append(low)
        ious.append(iou)
    low = np.concatenate(lows, 0)
    up = bilinear_resize(low, (1024, 1024))
    return {"low_res_logits": low, "iou_predictions": np.concatenate(ious, 0), "masks": up > 0.0, **f}


def greedy_generate(P: Params, llm_cfg: dict, inputs_embeds: np.ndarray, attention_mask: Optional[np.ndarray],
                    max_new_tokens: int, eos_token_id: int = EOS_TOKEN_ID, prefix: str = "language_model.") -> np.ndarray:
    """Greedy loop reproducing generate() modeling_internvl_sam.py:394-442 over
    prepare_inputs_for_generation modeling_internlm2.py:1112-1149: first step inputs_embeds, later steps
    the last id with position_ids = cumsum(mask)-1; stop at eos.  B=1. Returns new token ids."""
    assert inputs_embeds.shape[0] == 1
    S = inputs_embeds.shape[1]
    mask = np.ones((1, S), np.int64) if attention_mask is None else attention_mask.astype(np.int64)
    pos = np.cumsum(mask, -1) - 1
    pos[mask == 0] = 1
    h, past = internlm2_model(P, llm_cfg, inputs_embeds, mask, pos, None, True, prefix)
    out: List[int] = []
    for _ in range(max_new_tokens):
        tok = int(np.argmax(lm_head(P, h[:, -1:], prefix)[0, 0]))
        out.append(tok)
        if tok == eos_token_id:
            break
        mask = np.concatenate([mask, np.ones((1, 1), np.int64)], 1)
        pos = (np.cumsum(mask, -1) - 1)[:, -1:]
        e = P[prefix + "model.tok_embeddings.weight"][np.asarray([[tok]])].astype(F32)
        h, past = internlm2_model(P, llm_cfg, e, mask, pos, past, True, prefix)
    return np.asarray(out, np.int64)


def make_input_ids(n_text_pre: int = 20, n_text_post: int = 5, n_img: int = 1024, seed: int = 1, batch: int = 1) -> np.ndarray:
    """Synthetic ids (the tokenizer cannot be loaded): bos, text, <img>=92544,
    n_img x <IMG_CONTEXT>=92546, </img>=92545, text.  Default S = 1 + 20 + 1 + 1024 + 1 + 5 ... callers
    pick n_text_* to reach S=1081."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(batch):
        pre = rng.integers(3, 92000, n_text_pre)
        post = rng.integers(3, 92000, n_text_post)
        rows.append(np.concatenate([[1], pre, [92544], np.full(n_img, IMG_CONTEXT_TOKEN_ID), [92545], post]))
    return np.asarray(rows, np.int64)
