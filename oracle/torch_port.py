"""The oracle's two heavy stages -- a SAM ViT block and an InternLM2 decoder layer -- restated on torch CPU tensors (ATen kernels), for bench.py's
`cpu_baseline` only (kind "port-torch").  TEST / BASELINE INFRASTRUCTURE: nothing under ullsam_amd/ imports this file.

Why a second port: the reference's own CPU path runs on ATen (torch 2.5 there, 2.10 here), whose fused softmax / GELU / layer-norm kernels and oneDNN
matmuls are several times faster than the numpy restatement's chains of temporaries (round-4 review: ViT-H 53.6 s through numpy on 128 threads, where the
reference itself took 19.4 s on 8).  The arithmetic is the oracle's, function for function (each cites the same reference lines), and
tests/test_oracle_golden.py checks the two agree; the numpy oracle stays the parity checker."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(P, k):
    v = P[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def to_torch(P):
    """numpy parameter dict -> torch CPU tensors (shared memory)."""
    return {k: _t(P, k) for k in P}


def _rel_table(q_size: int, table: torch.Tensor) -> torch.Tensor:
    """get_rel_pos, image_encoder.py:292-322 (table lengths as built: no interpolation)."""
    i = torch.arange(q_size)
    return table[(i[:, None] - i[None, :]) + (q_size - 1)]


def vit_block(x: torch.Tensor, P, pre: str, num_heads: int, window_size: int, ln_eps: float) -> torch.Tensor:
    """Block.forward image_encoder.py:166-182 with window_partition / unpartition :243-289, Attention.forward :224-240, add_decomposed_rel_pos :325-361 (unscaled q),
    MLPBlock common.py:13-26.  x [B, H, W, D] fp32."""
    B, H, W, D = x.shape
    hd = D // num_heads
    h = F.layer_norm(x, (D,), P[pre + "norm1.weight"], P[pre + "norm1.bias"], ln_eps)
    if window_size > 0:
        ws = window_size
        ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
        h = F.pad(h, (0, 0, 0, pw, 0, ph))                                     # zero pad bottom / right AFTER norm1: pad tokens are live keys
        Hp, Wp = H + ph, W + pw
        h = h.reshape(B, Hp // ws, ws, Wp // ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, D)
        n = ws
    else:
        n = H
    Bp = h.shape[0]
    qkv = F.linear(h.reshape(Bp, n * n, D), P[pre + "attn.qkv.weight"], P[pre + "attn.qkv.bias"])
    q, k, v = qkv.reshape(Bp, n * n, 3, num_heads, hd).permute(2, 0, 3, 1, 4).unbind(0)      # [B', heads, N, hd]
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    rq = q.reshape(Bp, num_heads, n, n, hd)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, _rel_table(n, P[pre + "attn.rel_pos_h"]))
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, _rel_table(n, P[pre + "attn.rel_pos_w"]))
    attn = (attn.reshape(Bp, num_heads, n, n, n, n) + rel_h[..., :, None] + rel_w[..., None, :]).reshape(Bp, num_heads, n * n, n * n)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(Bp, n, n, D)
    o = F.linear(o, P[pre + "attn.proj.weight"], P[pre + "attn.proj.bias"])
    if window_size > 0:
        o = o.reshape(B, Hp // ws, Wp // ws, ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)[:, :H, :W]
    x = x + o
    m = F.layer_norm(x, (D,), P[pre + "norm2.weight"], P[pre + "norm2.bias"], ln_eps)
    m = F.linear(F.gelu(F.linear(m, P[pre + "mlp.lin1.weight"], P[pre + "mlp.lin1.bias"])), P[pre + "mlp.lin2.weight"], P[pre + "mlp.lin2.bias"])
    return x + m


def internlm2_layer(x: torch.Tensor, P, pre: str, cfg: dict) -> torch.Tensor:
    """InternLM2DecoderLayer.forward modeling_internlm2.py:621-681 at prefill without padding: RMSNorm :138-143, attention :341-426 (wqkv rearrange
    'b q (h gs d)', RoPE :233-247 with the tables of :147-180, the additive finfo.min causal mask of :96-110, fp32 softmax), SwiGLU MLP :261-264.  x [B, S, D] fp32."""
    B, S, D = x.shape
    H, KV = cfg["heads"], cfg["kv_heads"]
    hd, g = D // H, H // KV

    def rms(t, w):
        return w * (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + cfg["eps"]))

    h = rms(x, P[pre + "attention_norm.weight"])
    qkv = F.linear(h, P[pre + "attention.wqkv.weight"]).reshape(B, S, KV, g + 2, hd)
    q = qkv[..., :g, :].reshape(B, S, H, hd).transpose(1, 2)
    k = qkv[..., -2, :].transpose(1, 2)
    v = qkv[..., -1, :].transpose(1, 2)
    inv = 1.0 / (cfg["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]

    def rot(t):
        return torch.cat([-t[..., hd // 2:], t[..., :hd // 2]], -1)

    q = q * cos + rot(q) * sin
    k = k * cos + rot(k) * sin
    k = k[:, :, None].expand(B, KV, g, S, hd).reshape(B, H, S, hd)            # repeat_kv :268-277
    v = v[:, :, None].expand(B, KV, g, S, hd).reshape(B, H, S, hd)
    a = q @ k.transpose(-2, -1) / math.sqrt(hd)
    mask = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)
    a = (a + mask).softmax(-1, dtype=torch.float32)
    o = (a @ v).transpose(1, 2).reshape(B, S, D)
    x = x + F.linear(o, P[pre + "attention.wo.weight"])
    n = rms(x, P[pre + "ffn_norm.weight"])
    f = F.linear(F.silu(F.linear(n, P[pre + "feed_forward.w1.weight"])) * F.linear(n, P[pre + "feed_forward.w3.weight"]), P[pre + "feed_forward.w2.weight"])
    return x + f
