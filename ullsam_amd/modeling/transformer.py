"""Two-way transformer of the SAM mask decoder on HIP kernels (parameters mirror modeling/transformer.py).

Token side (T = 5 + prompts tokens) runs in fp32 with one-wave-per-output kernels; the image side (4096 tokens per
prompt) uses the MFMA GEMM in the model dtype with an fp32 `keys` stream.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple, Type

import torch
from torch import nn

from .. import ops
from .common import LayerNorm, Linear, MLPBlock, Packed


class Attention(Packed):
    """transformer.py:187-242 (q/k/v/out projections, optional internal downsampling)."""

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1) -> None:
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = Linear(embedding_dim, self.internal_dim)
        self.k_proj = Linear(embedding_dim, self.internal_dim)
        self.v_proj = Linear(embedding_dim, self.internal_dim)
        self.out_proj = Linear(self.internal_dim, embedding_dim)

    @property
    def hd(self):
        return self.internal_dim // self.num_heads

    def tok(self, lin: Linear, x, act=ops.ACT_NONE, res=None):
        return lin.tok(x, act, res)

    def attend_tokens(self, q, k, v, P, Tq, Tk):
        """softmax(q k^T / sqrt(hd)) v with the scale applied after QK^T (transformer.py:233-235); fp32 [P*T, internal]."""
        H, hd, C = self.num_heads, self.hd, self.internal_dim
        st = lambda T: (T * C, C, hd)
        return ops.naive_attention(q, k, v, P, H, H, hd, Tq, Tk, st(Tq), st(Tk), st(Tk), st(Tq), 1.0 / math.sqrt(hd))


class TwoWayAttentionBlock(Packed):
    """transformer.py:111-184."""

    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2, skip_first_layer_pe: bool = False) -> None:
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = LayerNorm(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim, activation)
        self.norm3 = LayerNorm(embedding_dim)
        self.norm4 = LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe


def _token_to_image(at: Attention, queries, qpe, keys_pe_c, keys_c, P, T, N, dt, shared=False, kv_cache=None, q=None, project=True):
    """queries + attn(q=queries+pe, k=keys+pe, v=keys): token side fp32, image-side K/V projections on MFMA (K/V kept in the
    model dtype), attention streamed over the image keys.  `shared`: one key set [N, C] for all P prompts.
    q given: the q projection was made by the fused token kernel; project False: return the attention output [P*T, internal] before the out projection."""
    if q is None:
        q = at.tok(at.q_proj, ops.add_cast(queries, qpe, torch.float32))
    if kv_cache is not None and "K0" in kv_cache:
        K, V = kv_cache["K0"], kv_cache["V0"]
    else:
        if FUSED_KV and dt == torch.bfloat16 and at.k_proj.weight.shape == (128, 256) and keys_c.shape == keys_pe_c.shape:
            # both projections in one pass, weights resident in LDS (any row count: a record's outputs must not depend on what it is batched with)
            K, V = ops.kv_proj(keys_pe_c, keys_c, at.k_proj.w(dt), at.k_proj.b(), at.v_proj.w(dt), at.v_proj.b())
        else:
            K = ops.gemm(keys_pe_c, at.k_proj.w(dt), at.k_proj.b())
            V = ops.gemm(keys_c, at.v_proj.w(dt), at.v_proj.b())
        if kv_cache is not None:
            kv_cache["K0"], kv_cache["V0"] = K, V
    if at.num_heads == 8 and at.hd == 16 and T <= (16 if K.dtype == torch.bfloat16 else 8):
        a = ops.tok2img_attention(q, K, V, P, at.num_heads, at.hd, T, N, 1.0 / math.sqrt(at.hd), kv_shared=shared)
    else:
        H, hd, C = at.num_heads, at.hd, at.internal_dim
        kst = (0 if shared else N * C, C, hd)
        a = ops.naive_attention(q, K.float(), V.float(), P, H, H, hd, T, N, (T * C, C, hd), kst, kst, (T * C, C, hd), 1.0 / math.sqrt(hd))
    if not project:
        return a
    return at.tok(at.out_proj, a, res=queries)


FUSED_KV = os.environ.get("ULLSAM_FUSED_KV", "1") != "0"     # bf16, SAM's decoder dimensions: the image side's k and v projections of a token -> image attention as one launch (csrc/decoder.hip kv_proj_kernel)
FUSED_TOK = os.environ.get("ULLSAM_FUSED_TOK", "1") != "0"   # bf16: the token side of a block as two launches (csrc/dectok.hip) instead of ~18


FUSED_I2T = os.environ.get("ULLSAM_FUSED_I2T", "1") != "0"   # image -> token half of a block as one kernel (bf16, SAM's decoder dimensions, >= 1024 image tokens; whatever the number of prompts, so
#                    that a record's outputs do not depend on what it is batched with)


class TwoWayTransformer(Packed):
    def __init__(self, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2) -> None:
        super().__init__()
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, activation, attention_downsample_rate, skip_first_layer_pe=(i == 0))
            for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = LayerNorm(embedding_dim)

    @property
    def compute_dtype(self):
        return self.final_attn_token_to_image.k_proj.weight.dtype

    def forward_tokens(self, keys: torch.Tensor, key_pe: torch.Tensor, tokens: torch.Tensor, keys_in_compute_dtype: bool = False,
                       cache: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """keys fp32 [P, N, C] (image embedding + dense prompt, token-major), key_pe fp32 [N, C], tokens fp32 [P, T, C].
        Returns (queries [P,T,C], keys [P,N,C]) as TwoWayTransformer.forward transformer.py:62-108; with `keys_in_compute_dtype`
        the returned keys are in the model dtype (what the mask decoder's upscaling consumes) and the fp32 copy is never written.
        keys may be [1, N, C] with P > 1 prompts (one image, many prompts, prompt-independent dense embedding): until the first
        image -> token attention the image side is identical for every prompt, so its projections of layer 0 are computed once and
        broadcast (same values as the reference's repeat_interleave, 1/P of the work)."""
        dt = self.compute_dtype
        Pk, N, C = keys.shape
        P, T = tokens.shape[0], tokens.shape[1]
        assert Pk in (1, P), (keys.shape, tokens.shape)
        shared = Pk == 1 and P > 1
        keys = keys.reshape(Pk * N, C)
        qpe = tokens.reshape(P * T, C).contiguous()
        queries = qpe
        f32 = torch.float32
        cache = cache if shared else None                # (`cache`: see MaskDecoder.predict_masks_tokens -- the prompt-independent image side of layer 0, kept across calls)
        if cache is not None and "keys_pe_c" in cache:
            keys_pe_c, keys_c = cache["keys_pe_c"], cache["keys_c"]
        else:
            keys_pe_c = ops.add_cast(keys, key_pe, dt)   # keys + key_pe: k of token->image, q of image->token
            keys_c = ops.cast(keys, dt)                  # v of token->image
            if cache is not None:
                cache["keys_pe_c"], cache["keys_c"] = keys_pe_c, keys_c
        fused_tok = (FUSED_TOK and dt == torch.bfloat16 and C == 256 and T <= 16 and all(
            b.self_attn.num_heads == 8 and b.self_attn.internal_dim == 256 and b.cross_attn_token_to_image.internal_dim == 128
            and b.cross_attn_image_to_token.internal_dim == 128 and b.mlp.lin1.out_features == 2048 and b.mlp.act_code == ops.ACT_RELU for b in self.layers)
            and self.final_attn_token_to_image.internal_dim == 128)
        for li, blk in enumerate(self.layers):
            if not fused_tok:
                break
            # ---- the token side as two launches around the token -> image attention (csrc/dectok.hip; same operations, activations rounded to bf16 at each linear's door)
            queries, q_t = ops.dec_tok_attn(queries, qpe, blk.self_attn, blk.norm1, blk.cross_attn_token_to_image.q_proj, P, T, blk.skip_first_layer_pe)
            a = _token_to_image(blk.cross_attn_token_to_image, queries, qpe, keys_pe_c, keys_c, P, T, N, dt, shared, kv_cache=cache if (shared and li == 0) else None,
                                q=q_t, project=False)
            ia = blk.cross_attn_image_to_token
            queries, k_i, v_i = ops.dec_tok_mlp(queries, a, qpe, blk.cross_attn_token_to_image.out_proj, blk.norm2, blk.mlp, blk.norm3, ia.k_proj, ia.v_proj, P, T)
            last = li + 1 == len(self.layers)
            if FUSED_I2T and ia.num_heads == 8 and N >= 1024 and key_pe.numel() == N * C:
                keys, keys_c, keys_pe_c = ops.i2t_block(keys_pe_c, keys, ia.q_proj.w(dt), ia.q_proj.b(), k_i, v_i, ia.out_proj.w(dt), ia.out_proj.b(),
                                                        *blk.norm4.wb(), blk.norm4.eps, key_pe, P, T, N, 1.0 / math.sqrt(ia.hd), shared,
                                                        want_f32=not (last and keys_in_compute_dtype))
            else:
                Qi = ops.gemm(keys_pe_c, ia.q_proj.w(dt), ia.q_proj.b(), out_f32=True)
                a = ops.fewkeys_attention(Qi, k_i, v_i, P, ia.num_heads, ia.hd, N, T, 1.0 / math.sqrt(ia.hd), q_shared=shared)
                upd = ops.gemm(ops.cast(a, dt), ia.out_proj.w(dt), ia.out_proj.b(), residual=keys, out_f32=True, res_row_mod=N if shared else 0)
                keys, keys_c, keys_pe_c = ops.norm_fanout(upd, *blk.norm4.wb(), blk.norm4.eps, dt, key_pe, want_f32=not (last and keys_in_compute_dtype))
            shared = False
        if fused_tok:
            fa = self.final_attn_token_to_image
            _, q_t = ops.dec_tok_attn(queries, qpe, None, None, fa.q_proj, P, T, False, mode=1)
            a = _token_to_image(fa, queries, qpe, keys_pe_c, keys_c, P, T, N, dt, shared, q=q_t, project=False)
            queries = ops.dec_tok_mlp(queries, a, None, fa.out_proj, self.norm_final_attn, None, None, None, None, P, T)
            out_keys = keys_c if keys_in_compute_dtype else keys
            return queries.reshape(P, T, C), out_keys.reshape(-1, N, C)
        for li, blk in enumerate(self.layers):
            sa = blk.self_attn
            if blk.skip_first_layer_pe:  # no PE and NO residual (:157-158)
                q_in, res = queries, None
            else:
                q_in, res = ops.add_cast(queries, qpe, f32), queries
            a = sa.attend_tokens(sa.tok(sa.q_proj, q_in), sa.tok(sa.k_proj, q_in), sa.tok(sa.v_proj, queries), P, T, T)
            queries = ops.norm(sa.tok(sa.out_proj, a, res=res), *blk.norm1.wb(), blk.norm1.eps, f32)
            queries = ops.norm(_token_to_image(blk.cross_attn_token_to_image, queries, qpe, keys_pe_c, keys_c, P, T, N, dt, shared,
                                               kv_cache=cache if (shared and li == 0) else None),
                               *blk.norm2.wb(), blk.norm2.eps, f32)
            m = blk.mlp
            hmid = m.lin1.tok(queries, m.act_code)
            queries = ops.norm(m.lin2.tok(hmid, res=queries), *blk.norm3.wb(), blk.norm3.eps, f32)
            # image -> token: keys = norm4(keys + attn(q=keys+pe, k=queries+pe, v=queries))   (:176-182)
            ia = blk.cross_attn_image_to_token
            q_in = ops.add_cast(queries, qpe, f32)
            last = li + 1 == len(self.layers)
            if (FUSED_I2T and dt == torch.bfloat16 and C == 256 and ia.internal_dim == 128 and ia.num_heads == 8 and T <= 16 and N >= 1024
                    and key_pe.numel() == N * C):
                # many prompts: q projection, attention over the T tokens, output projection + residual and norm4's fan-out in ONE pass over the stream
                keys, keys_c, keys_pe_c = ops.i2t_block(keys_pe_c, keys, ia.q_proj.w(dt), ia.q_proj.b(), ia.tok(ia.k_proj, q_in), ia.tok(ia.v_proj, queries), ia.out_proj.w(dt), ia.out_proj.b(),
                                                        *blk.norm4.wb(), blk.norm4.eps, key_pe, P, T, N, 1.0 / math.sqrt(ia.hd), shared,
                                                        want_f32=not (last and keys_in_compute_dtype))
            else:
                Qi = ops.gemm(keys_pe_c, ia.q_proj.w(dt), ia.q_proj.b(), out_f32=True)
                a = ops.fewkeys_attention(Qi, ia.tok(ia.k_proj, q_in), ia.tok(ia.v_proj, queries), P, ia.num_heads, ia.hd, N, T, 1.0 / math.sqrt(ia.hd), q_shared=shared)
                upd = ops.gemm(ops.cast(a, dt), ia.out_proj.w(dt), ia.out_proj.b(), residual=keys, out_f32=True,
                               res_row_mod=N if shared else 0)
                # norm4 feeds the next block's residual (fp32), its v projection (model dtype) and its k / q projections (+pe, model dtype)
                keys, keys_c, keys_pe_c = ops.norm_fanout(upd, *blk.norm4.wb(), blk.norm4.eps, dt, key_pe,
                                                          want_f32=not (last and keys_in_compute_dtype))
            shared = False                               # from here on every prompt has its own image-side stream
        fa = self.final_attn_token_to_image
        queries = ops.norm(_token_to_image(fa, queries, qpe, keys_pe_c, keys_c, P, T, N, dt, shared),
                           *self.norm_final_attn.wb(), self.norm_final_attn.eps, f32)
        out_keys = keys_c if keys_in_compute_dtype else keys
        return queries.reshape(P, T, C), out_keys.reshape(-1, N, C)

    @torch.no_grad()
    def forward(self, image_embedding: torch.Tensor, image_pe: torch.Tensor, point_embedding: torch.Tensor):
        """Reference signature: image_embedding/image_pe [B,C,h,w], point_embedding [B,T,C] -> (queries, keys[B,hw,C])."""
        bs, c, h, w = image_embedding.shape
        keys = ops.transpose(image_embedding.float().contiguous().reshape(bs, c, h * w), bs, c, h * w)
        pe = ops.transpose(image_pe.float().contiguous().reshape(-1, c, h * w)[:1], 1, c, h * w).reshape(h * w, c)
        return self.forward_tokens(keys, pe, point_embedding.float().contiguous())
