"""Thin torch-tensor wrappers over the C ABI (include/ullsam_hip.h).

torch is plumbing only: device memory, the current HIP stream, dtype bookkeeping.  Every function here launches
HIP kernels from libullsam_hip.so; none falls back to torch math.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_GELU, ACT_RELU, ACT_SWIGLU = 0, 1, 2, 3
ACT_SPLITK_OK = 256   # ullsam_hip.h ULLSAM_ACT_SPLITK_OK
_DT = {torch.float32: 0, torch.bfloat16: 1}


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"ullsam_amd supports float32 and bfloat16 compute, got {dtype}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise _lib.UllsamError(f"{name} must live on the GPU (ullsam_amd has no CPU path)")
    if t.device.index != torch.cuda.current_device():
        # kernels launch on the current device's current stream: a tensor of another GPU would be dereferenced on the wrong one
        raise _lib.UllsamError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                               f"run the call under `with torch.cuda.device({t.device.index}):`")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


_WS = {}


def _gemm_workspace(device) -> torch.Tensor:
    """Caller-owned scratch for the GEMM's split-K forms (one 128 MiB buffer per device AND stream: launches on different
    streams may overlap; the ring kernel's split-K keeps up to 8 fp32 planes of the output there, e.g. 4 x 1081 x 4096 x 4 B = 71 MB)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(128 << 20, dtype=torch.uint8, device=device)
    return ws


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         act: int = ACT_NONE, out_f32: bool = False, out: Optional[torch.Tensor] = None, res_row_mod: int = 0, splitk_ok: bool = False) -> torch.Tensor:
    """out[M, N'] = act(a[M,K] @ w[N,K]^T + bias) + residual   (N' = N/2 for ACT_SWIGLU).  splitk_ok: a launch of few tiles under a long K may run as K ranges
    summed apart (ULLSAM_ACT_SPLITK_OK, include/ullsam_hip.h): not bit-equal to the one-launch kernels -- the training step's frozen linears only."""
    _chk(a, "a"); _chk(w, "w", a.dtype)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    n_out = N // 2 if act == ACT_SWIGLU else N
    odt = torch.float32 if out_f32 else a.dtype
    if out is None:
        out = torch.empty((M, n_out), dtype=odt, device=a.device)
    else:
        _chk(out, "out", odt)
        assert out.shape == (M, n_out)
    if bias is not None:
        _chk(bias, "bias", torch.float32)
    ldr = 0
    if residual is not None:
        _chk(residual, "residual", torch.float32)
        ldr = residual.shape[-1]
    ws = _gemm_workspace(a.device)
    _lib.call("ullsam_gemm", dt_code(a.dtype), a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), n_out, int(out_f32),
              _p(bias), _p(residual), ldr, res_row_mod, act | (ACT_SPLITK_OK if splitk_ok else 0), M, N, K, ws.data_ptr(), ws.numel(), _stream())
    return out


def rows_fp8(x: torch.Tensor, ln_w: Optional[torch.Tensor] = None, ln_b: Optional[torch.Tensor] = None, eps: float = 0.0):
    """(optional LayerNorm, then) per-row e4m3 quantisation: x fp32 / bf16 [rows, D] -> (uint8 [rows, D] holding e4m3 bytes, fp32 scales [rows])."""
    _chk(x, "x")
    D = x.shape[-1]
    rows = x.numel() // D
    q = torch.empty((rows, D), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows,), dtype=torch.float32, device=x.device)
    if ln_w is not None:
        _chk(ln_w, "ln_w", torch.float32); _chk(ln_b, "ln_b", torch.float32)
    _lib.call("ullsam_rows_fp8", x.data_ptr(), dt_code(x.dtype), D, q.data_ptr(), D, sc.data_ptr(), _p(ln_w), _p(ln_b), rows, D, float(eps), _stream())
    return q, sc


def gemm_fp8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None,
             act: int = ACT_NONE, out_dtype: torch.dtype = torch.bfloat16, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = act((a8 @ w8^T) * a_scale[:, None] * w_scale[None, :] + bias) (+ residual); a8 / w8 are e4m3 bytes."""
    _chk(a8, "a8", torch.uint8); _chk(w8, "w8", torch.uint8); _chk(a_scale, "a_scale", torch.float32); _chk(w_scale, "w_scale", torch.float32)
    M, K = a8.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and a_scale.numel() == M and w_scale.numel() == N
    out_f32 = out_dtype == torch.float32
    out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    if bias is not None:
        _chk(bias, "bias", torch.float32)
    ldr = 0
    if residual is not None:
        _chk(residual, "residual", torch.float32)
        ldr = residual.shape[-1]
    _lib.call("ullsam_gemm_fp8", a8.data_ptr(), K, a_scale.data_ptr(), w8.data_ptr(), K, w_scale.data_ptr(), out.data_ptr(), N, int(out_f32),
              _p(bias), _p(residual), ldr, act, M, N, K, _stream())
    return out


def norm(x: torch.Tensor, w: Optional[torch.Tensor], b: Optional[torch.Tensor], eps: float, out_dtype: torch.dtype,
         rms: bool = False, act: int = ACT_NONE, post_scale: Optional[torch.Tensor] = None,
         post_shift: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, "x")
    D = x.shape[-1]
    rows = x.numel() // D
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    for t in (w, b, post_scale, post_shift):
        if t is not None:
            _chk(t, "norm param", torch.float32)
    _lib.call("ullsam_norm", x.data_ptr(), dt_code(x.dtype), D, out.data_ptr(), dt_code(out.dtype), D, _p(w), _p(b), rows, D,
              float(eps), int(rms), act, _p(post_scale), _p(post_shift), _stream())
    return out


def norm_fanout(x: torch.Tensor, w, b, eps: float, cdt: torch.dtype, pe: Optional[torch.Tensor], want_f32: bool = True,
                want_c: bool = True):
    """y = LayerNorm(x fp32 [rows, D]) -> (y fp32 | None, y in cdt | None, (y + pe[row % pe_rows]) in cdt | None) in one pass.
    With cdt == float32 the compute-dtype copy is the fp32 result itself."""
    _chk(x, "x", torch.float32)
    D = x.shape[-1]
    rows = x.numel() // D
    f32c = cdt == torch.float32
    out_f = torch.empty((rows, D), dtype=torch.float32, device=x.device) if (want_f32 or (f32c and want_c)) else None
    out_c = torch.empty((rows, D), dtype=cdt, device=x.device) if (want_c and not f32c) else None
    out_pe = torch.empty((rows, D), dtype=cdt, device=x.device) if pe is not None else None
    if pe is not None:
        _chk(pe, "pe", torch.float32)
    _lib.call("ullsam_norm_fanout", x.data_ptr(), rows, D, _p(w), _p(b), float(eps), _p(out_f), _p(out_c), _p(out_pe), dt_code(cdt),
              _p(pe), pe.numel() // D if pe is not None else 0, _stream())
    return out_f, (out_f if f32c else out_c), out_pe


def i2t_block(xin: torch.Tensor, res: torch.Tensor, wq: torch.Tensor, bq, ktok: torch.Tensor, vtok: torch.Tensor, wo: torch.Tensor, bo,
              lnw, lnb, eps: float, key_pe: Optional[torch.Tensor], P: int, T: int, N: int, scale: float, shared: bool, want_f32: bool = True,
              want_c: bool = True):
    """The image -> token half of a two-way block in one pass (csrc/decoder.hip i2t_block_kernel): xin = (keys + pe) bf16, res = keys fp32, both
    [P*N, 256] or [N, 256] when `shared`; ktok / vtok fp32 [P*T, 128]; -> (keys' fp32 | None, keys' bf16 | None, (keys' + pe) bf16 | None)."""
    _chk(xin, "xin", torch.bfloat16); _chk(res, "res", torch.float32); _chk(wq, "wq", torch.bfloat16); _chk(wo, "wo", torch.bfloat16)
    _chk(ktok, "ktok", torch.float32); _chk(vtok, "vtok", torch.float32)
    assert wq.shape == (128, 256) and wo.shape == (256, 128) and ktok.numel() == P * T * 128 and vtok.numel() == P * T * 128
    assert xin.numel() == (N if shared else P * N) * 256 and res.numel() == xin.numel()
    rows = P * N
    out_f = torch.empty((rows, 256), dtype=torch.float32, device=xin.device) if want_f32 else None
    out_c = torch.empty((rows, 256), dtype=torch.bfloat16, device=xin.device) if want_c else None
    out_pe = torch.empty((rows, 256), dtype=torch.bfloat16, device=xin.device) if key_pe is not None else None
    if key_pe is not None:
        _chk(key_pe, "key_pe", torch.float32)
    _lib.call("ullsam_i2t_block", xin.data_ptr(), N if shared else 0, res.data_ptr(), N if shared else 0, wq.data_ptr(), _p(bq), ktok.data_ptr(), vtok.data_ptr(),
              wo.data_ptr(), _p(bo), _p(lnw), _p(lnb), float(eps), _p(key_pe), key_pe.numel() // 256 if key_pe is not None else 0, _p(out_f), _p(out_c), _p(out_pe),
              P, T, N, float(scale), _stream())
    return out_f, out_c, out_pe


def kv_proj(xk: torch.Tensor, xv: torch.Tensor, wk: torch.Tensor, bk, wv: torch.Tensor, bv):
    """K = xk wk^T + bk, V = xv wv^T + bv in one pass over the image side (csrc/decoder.hip kv_proj_kernel): xk / xv bf16 [rows, 256], wk / wv bf16 [128, 256] -> bf16 [rows, 128] x 2."""
    _chk(xk, "xk", torch.bfloat16); _chk(xv, "xv", torch.bfloat16); _chk(wk, "wk", torch.bfloat16); _chk(wv, "wv", torch.bfloat16)
    rows = xk.numel() // 256
    assert xk.shape[-1] == 256 and xv.shape == xk.shape and wk.shape == (128, 256) and wv.shape == (128, 256)
    K = torch.empty((rows, 128), dtype=torch.bfloat16, device=xk.device)
    V = torch.empty((rows, 128), dtype=torch.bfloat16, device=xk.device)
    _lib.call("ullsam_kv_proj", xk.data_ptr(), xv.data_ptr(), wk.data_ptr(), wv.data_ptr(), _p(bk), _p(bv), K.data_ptr(), V.data_ptr(), rows, _stream())
    return K, V


def pack_mfma_rows(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear weight [out, in] (out padded to a multiple of 16 with zero rows; in % 32 == 0) -> bf16 in MFMA A-fragment order [out / 16][in / 32][lane = 16 g + m][8]:
    lane (m, g) of row tile t and k-step s holds w[16 t + m][32 s + 8 g .. + 7], and a wave's fragment load is one contiguous KiB (csrc/dectok.hip lin_tiles).
    A re-layout (reshape / permute / copy), done once per weight version by the callers' pack caches."""
    o, i = w.shape
    assert i % 32 == 0
    w = w.detach().to(torch.bfloat16)
    if o % 16:
        w = torch.cat([w, torch.zeros(((-o) % 16, i), dtype=w.dtype, device=w.device)], 0)
    return w.reshape(-1, 16, i // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def _fw(lin) -> torch.Tensor:
    return lin.pk("w:mfma_rows", lin.weight, lambda: pack_mfma_rows(lin.weight))


def dec_tok_attn(queries, qpe, sa, norm, q2_lin, P: int, T: int, skip_pe: bool, mode: int = 0):
    """Fused token-side self-attention half of a two-way block (csrc/dectok.hip): queries / qpe fp32 [P*T, 256]; sa = the block's self-attention module
    (q / k / v / out projections, bf16), norm = norm1, q2_lin = the token -> image attention's q projection.  -> (queries' fp32 [P*T, 256], q fp32 [P*T, 128]).
    mode 1: only the q projection of (queries + qpe) (the final attention): -> (queries, q)."""
    _chk(queries, "queries", torch.float32); _chk(qpe, "qpe", torch.float32)
    bf = torch.bfloat16
    q2 = torch.empty((P * T, 128), dtype=torch.float32, device=queries.device)
    if mode == 1:
        _lib.call("ullsam_dec_tok_attn", queries.data_ptr(), qpe.data_ptr(), None, q2.data_ptr(), None, None, None, None, None, None, None, None, None, None, 0.0,
                  _fw(q2_lin).data_ptr(), _p(q2_lin.b()), P, T, 0, 1, _stream())
        return queries, q2
    out = torch.empty_like(queries)
    lw, lb = norm.wb()
    _lib.call("ullsam_dec_tok_attn", queries.data_ptr(), qpe.data_ptr(), out.data_ptr(), q2.data_ptr(), _fw(sa.q_proj).data_ptr(), _p(sa.q_proj.b()),
              _fw(sa.k_proj).data_ptr(), _p(sa.k_proj.b()), _fw(sa.v_proj).data_ptr(), _p(sa.v_proj.b()), _fw(sa.out_proj).data_ptr(), _p(sa.out_proj.b()),
              lw.data_ptr(), lb.data_ptr(), float(norm.eps), _fw(q2_lin).data_ptr(), _p(q2_lin.b()), P, T, int(skip_pe), 0, _stream())
    return out, q2


def dec_tok_mlp(queries, attn, qpe, out_lin, norm2, mlp, norm3, k_lin, v_lin, P: int, T: int):
    """Fused second half of a block's token side (csrc/dectok.hip): queries fp32 [P*T, 256] (after norm1), attn fp32 [P*T, 128] (token -> image attention output) ->
    out projection + residual, norm2, MLP + residual, norm3, and the image -> token k / v projections: (queries' [P*T, 256], k [P*T, 128], v [P*T, 128]) fp32.
    mlp None: out projection + residual + norm2 only (the final attention with norm_final_attn): -> queries'."""
    _chk(queries, "queries", torch.float32); _chk(attn, "attn", torch.float32)
    bf = torch.bfloat16
    out = torch.empty_like(queries)
    w2, b2 = norm2.wb()
    if mlp is None:
        _lib.call("ullsam_dec_tok_mlp", queries.data_ptr(), attn.data_ptr(), None, out.data_ptr(), None, None, _fw(out_lin).data_ptr(), _p(out_lin.b()), w2.data_ptr(), b2.data_ptr(),
                  float(norm2.eps), None, None, None, None, None, None, 0.0, None, None, None, None, P, T, 0, _stream())
        return out
    _chk(qpe, "qpe", torch.float32)
    k = torch.empty((P * T, 128), dtype=torch.float32, device=queries.device)
    v = torch.empty_like(k)
    w3, b3 = norm3.wb()
    _lib.call("ullsam_dec_tok_mlp", queries.data_ptr(), attn.data_ptr(), qpe.data_ptr(), out.data_ptr(), k.data_ptr(), v.data_ptr(), _fw(out_lin).data_ptr(), _p(out_lin.b()),
              w2.data_ptr(), b2.data_ptr(), float(norm2.eps), _fw(mlp.lin1).data_ptr(), _p(mlp.lin1.b()), _fw(mlp.lin2).data_ptr(), _p(mlp.lin2.b()), w3.data_ptr(), b3.data_ptr(),
              float(norm3.eps), _fw(k_lin).data_ptr(), _p(k_lin.b()), _fw(v_lin).data_ptr(), _p(v_lin.b()), P, T, 1, _stream())
    return out, k, v


def dec_heads(hs: torch.Tensor, w_ptrs: torch.Tensor, b_ptrs: torch.Tensor, P: int, T: int, n_iou: int, m0: int = 0, nm: int = 4):
    """The hypernetwork MLPs of masks m0 .. m0 + nm - 1 + the IoU head in one launch (csrc/dectok.hip): hs fp32 [P, T, 256]; w_ptrs / b_ptrs = HOST int64
    tensors of 15 device pointers (chain-major, three layers each; built and kept alive by MaskDecoder).  -> (hyper fp32 [P, nm, 32], iou fp32 [P, n_iou])."""
    _chk(hs, "hs", torch.float32)
    hyper = torch.empty((P, nm, 32), dtype=torch.float32, device=hs.device)
    iou = torch.empty((P, n_iou), dtype=torch.float32, device=hs.device)
    _lib.call("ullsam_dec_heads", hs.data_ptr(), w_ptrs.data_ptr(), b_ptrs.data_ptr(), hyper.data_ptr(), iou.data_ptr(), P, T, n_iou, m0, nm, _stream())
    return hyper, iou


def concat_token_rows(prefix: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    """[n0, C] fp32 shared rows in front of [P, n1, C] fp32 per-prompt rows -> [P, n0 + n1, C]: the decoder's token matrix in one launch."""
    _chk(prefix, "prefix", torch.float32); _chk(rows, "rows", torch.float32)
    P, n1, C = rows.shape
    out = torch.empty((P, prefix.shape[0] + n1, C), dtype=torch.float32, device=rows.device)
    _lib.call("ullsam_concat_token_rows", prefix.data_ptr(), prefix.shape[0], rows.data_ptr() if n1 else None, n1, out.data_ptr(), P, C, _stream())
    return out


def vit_attention(qkv: torch.Tensor, rel_h: torch.Tensor, rel_w: torch.Tensor, qkv_bias: torch.Tensor, B: int, heads: int,
                  hd: int, gh: int, gw: int, window: int) -> torch.Tensor:
    _chk(qkv, "qkv"); _chk(rel_h, "rel_h", qkv.dtype); _chk(rel_w, "rel_w", qkv.dtype); _chk(qkv_bias, "qkv_bias", qkv.dtype)
    D = heads * hd
    assert qkv.numel() == B * gh * gw * 3 * D
    n = window if window > 0 else gh
    assert rel_h.shape == (2 * n - 1, hd) and rel_w.shape[1] == hd
    out = torch.empty((B * gh * gw, D), dtype=qkv.dtype, device=qkv.device)
    _lib.call("ullsam_vit_attention", dt_code(qkv.dtype), qkv.data_ptr(), out.data_ptr(), rel_h.data_ptr(), rel_w.data_ptr(),
              qkv_bias.data_ptr(), B, heads, hd, gh, gw, window, _stream())
    return out


def causal_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, key_mask: Optional[torch.Tensor],
                     B: int, H: int, KVH: int, hd: int, Sq: int, Sk: int, q_pos0: int) -> torch.Tensor:
    _chk(q, "q"); _chk(k_cache, "k_cache", q.dtype); _chk(v_cache, "v_cache", q.dtype)
    cap = k_cache.shape[2]
    if key_mask is not None:
        _chk(key_mask, "key_mask", torch.int32)
        assert key_mask.shape == (B, Sk)
    out = torch.empty((B * Sq, H * hd), dtype=q.dtype, device=q.device)
    _lib.call("ullsam_causal_attention", dt_code(q.dtype), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
              _p(key_mask), B, H, KVH, hd, Sq, Sk, cap, q_pos0, _stream())
    return out


def naive_attention(q, k, v, B, H, KVH, hd, Sq, Sk, qs, ks, vs, os_, scale, key_mask=None, out=None):
    """Generic strided attention; qs/ks/vs/os_ = (batch, token, head) element strides."""
    _chk(q, "q"); _chk(k, "k", q.dtype); _chk(v, "v", q.dtype)
    if out is None:
        out = torch.empty((B * Sq, H * hd), dtype=q.dtype, device=q.device)
    if key_mask is not None:
        _chk(key_mask, "key_mask", torch.int32)
    _lib.call("ullsam_naive_attention", dt_code(q.dtype), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _p(key_mask),
              B, H, KVH, hd, Sq, Sk, *qs, *ks, *vs, *os_, float(scale), _stream())
    return out


def fewkeys_attention(q, k, v, B, H, hd, Sq, Sk, scale, q_shared: bool = False):
    """q fp32 [B (or 1 when q_shared), Sq, H*hd], k/v fp32 [B, Sk, H*hd] -> fp32 [B*Sq, H*hd]."""
    for t in (q, k, v):
        _chk(t, "qkv", torch.float32)
    out = torch.empty((B * Sq, H * hd), dtype=torch.float32, device=q.device)
    _lib.call("ullsam_fewkeys_attention", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, hd, Sq, Sk,
              float(scale), 0 if q_shared else Sq * H * hd, _stream())
    return out


def decode_attention(q, kc, vc, key_mask, B, H, KVH, hd, Sk):
    """q bf16 [B, H*hd] (one new token per sequence), kc / vc bf16 caches [B, KVH, cap, hd], key_mask int32 [B, Sk] | None -> bf16 [B, H*hd]."""
    _chk(q, "q", torch.bfloat16); _chk(kc, "k cache", torch.bfloat16); _chk(vc, "v cache", torch.bfloat16)
    if key_mask is not None:
        _chk(key_mask, "key_mask", torch.int32)
    G, P = H // KVH, B * KVH
    nsplit = max(1, min(32, -(-256 // P), Sk // 64))     # one workgroup per CU (8 splits at batch 4: 3.605 ms per step against 3.639 with 16, 3.63 with 4)
    ws = torch.empty((P * nsplit * G * (hd + 2),), dtype=torch.float32, device=q.device)
    out = torch.empty((B, H * hd), dtype=torch.bfloat16, device=q.device)
    _lib.call("ullsam_decode_attention", q.data_ptr(), kc.data_ptr(), vc.data_ptr(), _p(key_mask), out.data_ptr(), B, H, KVH, hd, Sk,
              kc.shape[2], float(hd) ** -0.5, ws.data_ptr(), nsplit, _stream())
    return out


def tok2img_attention(q, k, v, P, H, hd, T, N, scale, kv_shared: bool = False):
    """q fp32 [P*T, H*hd]; k, v [P (or 1 when kv_shared) * N, H*hd] fp32 or bf16 -> fp32 [P*T, H*hd]."""
    _chk(q, "q", torch.float32); _chk(k, "k"); _chk(v, "v", k.dtype)
    C = H * hd
    nsplit = max(1, min(8, N // 128))          # independent of P: the partition of the keys (and with it the order of the softmax merge) must not depend on how many prompts share the launch
    ws = torch.empty((P * nsplit * T * (C + 2 * H),), dtype=torch.float32, device=q.device)
    out = torch.empty((P * T, C), dtype=torch.float32, device=q.device)
    bs = 0 if kv_shared else N * C
    _lib.call("ullsam_tok2img_attention", dt_code(k.dtype), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), P, H, hd, T, N,
              bs, bs, float(scale), ws.data_ptr(), nsplit, _stream())
    return out


def patch_im2col(pixels: torch.Tensor, S: int, patch: int, dtype: torch.dtype, mean=None, std=None) -> torch.Tensor:
    _chk(pixels, "pixels", torch.float32)
    B, C, Hs, Ws = pixels.shape
    g = S // patch
    out = torch.empty((B * g * g, C * patch * patch), dtype=dtype, device=pixels.device)
    _lib.call("ullsam_patch_im2col", dt_code(dtype), pixels.data_ptr(), out.data_ptr(), B, C, Hs, Ws, S, patch, _p(mean), _p(std),
              _stream())
    return out


def im2col3x3(x: torch.Tensor, B: int, H: int, W: int, C: int) -> torch.Tensor:
    _chk(x, "x")
    out = torch.empty((B * H * W, 9 * C), dtype=x.dtype, device=x.device)
    _lib.call("ullsam_im2col3x3", dt_code(x.dtype), x.data_ptr(), out.data_ptr(), B, H, W, C, _stream())
    return out


def add_cast(a: torch.Tensor, b: Optional[torch.Tensor], out_dtype: torch.dtype, rows: Optional[int] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = a[r % a_rows] + b[r % b_rows] (b optional, fp32), converted to out_dtype."""
    _chk(a, "a")
    cols = a.shape[-1]
    a_rows = a.numel() // cols
    b_rows = 0
    if b is not None:
        _chk(b, "b", torch.float32)
        assert b.shape[-1] == cols
        b_rows = b.numel() // cols
    if rows is None:
        rows = max(a_rows, b_rows)
    if out is None:
        out = torch.empty((rows, cols), dtype=out_dtype, device=a.device)
    _lib.call("ullsam_add_cast", a.data_ptr(), dt_code(a.dtype), a_rows, _p(b), b_rows, out.data_ptr(), dt_code(out.dtype), rows,
              cols, _stream())
    return out


def cast(a: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    if a.dtype == out_dtype:
        return a
    return add_cast(a.reshape(-1, a.shape[-1]), None, out_dtype).reshape(a.shape)


def transpose(x: torch.Tensor, B: int, R: int, C: int) -> torch.Tensor:
    """fp32 [B, R, C] -> [B, C, R]."""
    _chk(x, "x", torch.float32)
    out = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
    _lib.call("ullsam_transpose_f32", x.data_ptr(), out.data_ptr(), B, R, C, _stream())
    return out


def transpose_to_bf16(x: torch.Tensor, pad_to: int = 1) -> torch.Tensor:
    """fp32 / bf16 [R, C] -> bf16 [C, R rounded up to pad_to] (zero filled), cast and transposed in one pass."""
    _chk(x, "x")
    R, Cn = x.shape
    Rp = -(-R // pad_to) * pad_to
    out = torch.empty((Cn, Rp), dtype=torch.bfloat16, device=x.device)
    _lib.call("ullsam_transpose_to_bf16", dt_code(x.dtype), x.data_ptr(), out.data_ptr(), R, Cn, Rp, _stream())
    return out


def cast_transpose_bf16(x: torch.Tensor, pad_to: int = 64, row_major: bool = True, colsum: bool = False):
    """fp32 [R, C] -> (bf16 [R, C] | None, bf16 [C, R rounded up to pad_to] zero filled, fp32 [C] column sums | None) in one pass over x (csrc/vit_misc.hip
    cast_transpose_bf16_kernel); C % 4 == 0 and pad_to % 4 == 0."""
    _chk(x, "x", torch.float32)
    R, Cn = x.shape
    assert Cn % 4 == 0 and pad_to % 4 == 0
    Rp = -(-R // pad_to) * pad_to
    rm = torch.empty((R, Cn), dtype=torch.bfloat16, device=x.device) if row_major else None
    xt = torch.empty((Cn, Rp), dtype=torch.bfloat16, device=x.device)
    cs = ws = None
    if colsum:
        cs = torch.empty((Cn,), dtype=torch.float32, device=x.device)
        ws = torch.empty((-(-Rp // 64) * Cn,), dtype=torch.float32, device=x.device)
    _lib.call("ullsam_cast_transpose_bf16", x.data_ptr(), _p(rm), xt.data_ptr(), _p(cs), _p(ws), R, Cn, Rp, _stream())
    return rm, xt, cs


def pixel_shuffle_ln(x_nhwc: torch.Tensor, w, b, B, H, W, C, eps, dtype) -> torch.Tensor:
    _chk(x_nhwc, "x", torch.float32)
    out = torch.empty((B * (H // 2) * (W // 2), 4 * C), dtype=dtype, device=x_nhwc.device)
    _lib.call("ullsam_pixel_shuffle_ln", dt_code(dtype), x_nhwc.data_ptr(), out.data_ptr(), w.data_ptr(), b.data_ptr(), B, H, W, C,
              float(eps), _stream())
    return out


def pixel_unshuffle(x: torch.Tensor, B, H, W, C) -> torch.Tensor:
    _chk(x, "x", torch.float32)
    out = torch.empty((B, H * W, C), dtype=torch.float32, device=x.device)
    _lib.call("ullsam_pixel_unshuffle", x.data_ptr(), out.data_ptr(), B, H, W, C, _stream())
    return out


def scan_image_tokens(ids: torch.Tensor, img_id: int):
    _chk(ids, "input_ids", torch.int64)
    B, S = ids.shape
    rank = torch.empty((B, S), dtype=torch.int32, device=ids.device)
    rng = torch.empty((B, 2), dtype=torch.int32, device=ids.device)
    _lib.call("ullsam_scan_image_tokens", ids.data_ptr(), rank.data_ptr(), rng.data_ptr(), B, S, int(img_id), _stream())
    return rank, rng


def embed_tokens(table: torch.Tensor, ids: torch.Tensor, rank: Optional[torch.Tensor], vit: Optional[torch.Tensor]) -> torch.Tensor:
    _chk(table, "tok_embeddings"); _chk(ids, "ids", torch.int64)
    B, S = ids.shape
    V, D = table.shape
    n_img = 0
    if vit is not None:
        _chk(vit, "vit_embeds", torch.float32)
        n_img = vit.numel() // (B * D)
    out = torch.empty((B * S, D), dtype=torch.float32, device=ids.device)
    _lib.call("ullsam_embed_tokens", dt_code(table.dtype), table.data_ptr(), ids.data_ptr(), _p(rank), _p(vit), out.data_ptr(), B, S, D,
              max(n_img, 1), V, _stream())
    return out


def gather_rows(x: torch.Tensor, rng: torch.Tensor, B: int, S: int, n: int) -> torch.Tensor:
    _chk(x, "x"); _chk(rng, "range", torch.int32)
    D = x.shape[-1]
    out = torch.empty((B * n, D), dtype=x.dtype, device=x.device)
    _lib.call("ullsam_gather_rows", x.data_ptr(), out.data_ptr(), rng.data_ptr(), B, S, n, D * x.element_size(), _stream())
    return out


def rope_split(qkv, k_cache, v_cache, pos, cos_tab, sin_tab, B, S, KVH, G, hd, cache_pos0) -> torch.Tensor:
    _chk(qkv, "qkv"); _chk(pos, "position_ids", torch.int32); _chk(cos_tab, "cos", torch.float32); _chk(sin_tab, "sin", torch.float32)
    q = torch.empty((B * S, KVH * G * hd), dtype=qkv.dtype, device=qkv.device)
    _lib.call("ullsam_rope_split", dt_code(qkv.dtype), qkv.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
              pos.data_ptr(), cos_tab.data_ptr(), sin_tab.data_ptr(), B, S, KVH, G, hd, k_cache.shape[2], cache_pos0, cos_tab.shape[0], _stream())
    return q


def gemm_qkv_rope(x: torch.Tensor, wqkv: torch.Tensor, bias: Optional[torch.Tensor], k_cache, v_cache, pos, cos_tab, sin_tab, B, S, KVH, G,
                  cache_pos0) -> torch.Tensor:
    """q = rope(split(x @ wqkv^T + bias)) with k (rotated) / v appended to the caches, all in the GEMM's epilogue (head_dim 128)."""
    _chk(x, "x"); _chk(wqkv, "wqkv", x.dtype); _chk(pos, "position_ids", torch.int32); _chk(cos_tab, "cos", torch.float32); _chk(sin_tab, "sin", torch.float32)
    _chk(k_cache, "k_cache", x.dtype); _chk(v_cache, "v_cache", x.dtype)
    K = x.shape[1]
    assert wqkv.shape == (KVH * (G + 2) * 128, K) and x.shape[0] == B * S and cos_tab.shape[1] == 128
    if bias is not None:
        _chk(bias, "bias", torch.float32)
    q = torch.empty((B * S, KVH * G * 128), dtype=x.dtype, device=x.device)
    ws = _gemm_workspace(x.device)
    _lib.call("ullsam_gemm_qkv_rope", dt_code(x.dtype), x.data_ptr(), K, wqkv.data_ptr(), K, _p(bias), B, S, K, KVH, G, pos.data_ptr(),
              cos_tab.data_ptr(), sin_tab.data_ptr(), cos_tab.shape[0], q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_cache.shape[2],
              cache_pos0, ws.data_ptr(), ws.numel(), _stream())
    return q


def decode_fusable(M: int, K: int, dtype: torch.dtype) -> bool:
    """Shapes the decode-step kernels with the fused RMSNorm prologue / RoPE epilogue take (csrc/gemm.hip launch_gemm_skinny)."""
    return dtype == torch.bfloat16 and 1 <= M <= 4 and K <= 4096 and K % 2048 == 0


def gemm_rmsnorm(x: torch.Tensor, norm_w: torch.Tensor, eps: float, w: torch.Tensor, act: int = ACT_NONE) -> torch.Tensor:
    """act(bf16(RMSNorm(x) * norm_w) @ w^T) for a decode step's M <= 4 fp32 rows: the norm is computed while the weight stream starts."""
    _chk(x, "x", torch.float32); _chk(norm_w, "norm_w", torch.float32); _chk(w, "w", torch.bfloat16)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and decode_fusable(M, K, w.dtype), (x.shape, w.shape)
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty((M, n_out), dtype=torch.bfloat16, device=x.device)
    _lib.call("ullsam_gemm_rmsnorm", x.data_ptr(), K, norm_w.data_ptr(), float(eps), w.data_ptr(), K, out.data_ptr(), n_out, 0, None, None, 0,
              act, M, N, K, _stream())
    return out


def decode_qkv_rope(x: torch.Tensor, norm_w: Optional[torch.Tensor], eps: float, wqkv: torch.Tensor, bias: Optional[torch.Tensor], k_cache, v_cache,
                    pos, cos_tab, sin_tab, B, KVH, G, cache_pos0) -> torch.Tensor:
    """One new token per sequence (B <= 4): q = rope(split(h @ wqkv^T + bias)), k (rotated) / v appended to the caches at cache_pos0, where
    h = bf16(RMSNorm(x) * norm_w) for fp32 x (norm_w given) or x itself (bf16, norm_w None)."""
    _chk(wqkv, "wqkv", torch.bfloat16); _chk(pos, "position_ids", torch.int32); _chk(cos_tab, "cos", torch.float32); _chk(sin_tab, "sin", torch.float32)
    _chk(k_cache, "k_cache", torch.bfloat16); _chk(v_cache, "v_cache", torch.bfloat16)
    _chk(x, "x", torch.float32 if norm_w is not None else torch.bfloat16)
    K = x.shape[1]
    assert wqkv.shape == (KVH * (G + 2) * 128, K) and x.shape[0] == B and cos_tab.shape[1] == 128 and decode_fusable(B, K, wqkv.dtype)
    if bias is not None:
        _chk(bias, "bias", torch.float32)
    if norm_w is not None:
        _chk(norm_w, "norm_w", torch.float32)
    q = torch.empty((B, KVH * G * 128), dtype=torch.bfloat16, device=x.device)
    _lib.call("ullsam_decode_qkv_rope", None if norm_w is not None else x.data_ptr(), x.data_ptr() if norm_w is not None else None, K, _p(norm_w),
              float(eps), wqkv.data_ptr(), K, _p(bias), B, K, KVH, G, pos.data_ptr(), cos_tab.data_ptr(), sin_tab.data_ptr(), cos_tab.shape[0],
              q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_cache.shape[2], cache_pos0, _stream())
    return q


def argmax(logits: torch.Tensor) -> torch.Tensor:
    _chk(logits, "logits", torch.float32)
    R, V = logits.shape
    out = torch.empty((R,), dtype=torch.int64, device=logits.device)
    _lib.call("ullsam_argmax", logits.data_ptr(), out.data_ptr(), R, V, V, _stream())
    return out


def small_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int = ACT_NONE,
                 res: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, "x", torch.float32); _chk(w, "w", torch.float32)
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.call("ullsam_small_linear", x.data_ptr(), K, w.data_ptr(), _p(b), _p(res), N, out.data_ptr(), N, M, N, K, act, _stream())
    return out


def skinny_linear(x: torch.Tensor, wt: torch.Tensor, b: Optional[torch.Tensor], act: int = ACT_NONE,
                  res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 Linear for tens..hundreds of rows; wt = weight^T [K, N] fp32."""
    _chk(x, "x", torch.float32); _chk(wt, "wt", torch.float32)
    K, N = wt.shape
    M = x.numel() // K
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.call("ullsam_skinny_linear", x.data_ptr(), K, wt.data_ptr(), _p(b), _p(res), N, out.data_ptr(), N, M, N, K, act, _stream())
    return out


def sparse_embed(coords, labels, boxes, G, emb, P, Np, pad, C, img_w, img_h) -> torch.Tensor:
    n_out = Np + pad + (2 if boxes is not None else 0)
    out = torch.empty((P, n_out, C), dtype=torch.float32, device=G.device)
    _lib.call("ullsam_sparse_embed", _p(coords), _p(labels), _p(boxes), G.data_ptr(), emb.data_ptr(), out.data_ptr(), P, Np, pad, C,
              float(img_w), float(img_h), _stream())
    return out


def dense_pe(G: torch.Tensor, H: int, W: int) -> torch.Tensor:
    C = G.shape[1] * 2
    out = torch.empty((H * W, C), dtype=torch.float32, device=G.device)
    _lib.call("ullsam_dense_pe", G.data_ptr(), out.data_ptr(), H, W, C, _stream())
    return out


def mask_downscale(masks: torch.Tensor, H: int, W: int, C: int, params) -> torch.Tensor:
    _chk(masks, "masks", torch.float32)
    P = masks.shape[0]
    c1, c2 = params[0].shape[0], params[4].shape[0]
    out = torch.empty((P, H * W, C), dtype=torch.float32, device=masks.device)
    _lib.call("ullsam_mask_downscale", masks.data_ptr(), out.data_ptr(), P, H, W, C, c1, c2, *[t.data_ptr() for t in params], _stream())
    return out


def hyper_masks(up2: torch.Tensor, hyper: torch.Tensor, NB: int, NM: int, H: int, W: int, CU: int) -> torch.Tensor:
    out = torch.empty((NB, NM, 4 * H, 4 * W), dtype=torch.float32, device=up2.device)
    _lib.call("ullsam_hyper_masks", dt_code(up2.dtype), up2.data_ptr(), hyper.data_ptr(), out.data_ptr(), NB, NM, H, W, CU, _stream())
    return out


def up1_ln_gelu(src: torch.Tensor, w0: torch.Tensor, b0, lnw, lnb, eps: float) -> torch.Tensor:
    """First transposed convolution (as Linear 256 -> 4 x 64) + LayerNorm2d + GELU in one pass (bf16): src [rows, 256], w0 [256, 256] -> bf16 [rows * 4, 64]."""
    _chk(src, "src", torch.bfloat16); _chk(w0, "w0", torch.bfloat16)
    rows = src.shape[0]
    assert src.shape == (rows, 256) and w0.shape == (256, 256)
    out = torch.empty((rows * 4, 64), dtype=torch.bfloat16, device=src.device)
    _lib.call("ullsam_up1_ln_gelu", src.data_ptr(), w0.data_ptr(), _p(b0), _p(lnw), _p(lnb), float(eps), out.data_ptr(), rows, _stream())
    return out


def up2_hyper_masks(u1: torch.Tensor, w1: torch.Tensor, b1, hyper: torch.Tensor, NB: int, NM: int, H: int, W: int) -> torch.Tensor:
    """Second transposed convolution (as Linear 64 -> 4 x 32) + GELU + hypernetwork product in one pass (bf16): u1 [NB*H*W*4, 64], w1 [128, 64], hyper fp32
    [NB, NM, 32] -> fp32 [NB, NM, 4H, 4W]."""
    _chk(u1, "u1", torch.bfloat16); _chk(w1, "w1", torch.bfloat16); _chk(hyper, "hyper", torch.float32)
    assert u1.shape == (NB * H * W * 4, 64) and w1.shape == (128, 64) and hyper.numel() == NB * NM * 32
    out = torch.empty((NB, NM, 4 * H, 4 * W), dtype=torch.float32, device=u1.device)
    _lib.call("ullsam_up2_hyper_masks", u1.data_ptr(), w1.data_ptr(), _p(b1), hyper.data_ptr(), out.data_ptr(), NB, NM, H, W, _stream())
    return out


def resize_bilinear(x: torch.Tensor, out_hw, valid_hw=None, want_float=True, threshold: Optional[float] = None):
    """x fp32 [..., IH, IW] (optionally only the top-left valid_hw region is the source image)."""
    _chk(x, "x", torch.float32)
    IHs, IWs = x.shape[-2:]
    N = x.numel() // (IHs * IWs)
    IH, IW = valid_hw if valid_hw is not None else (IHs, IWs)
    OH, OW = out_hw
    out = torch.empty(x.shape[:-2] + (OH, OW), dtype=torch.float32, device=x.device) if want_float else None
    mask = torch.empty(x.shape[:-2] + (OH, OW), dtype=torch.uint8, device=x.device) if threshold is not None else None
    _lib.call("ullsam_resize_bilinear", x.data_ptr(), IHs * IWs, IWs, IH, IW, _p(out), _p(mask), N, OH, OW,
              float(threshold if threshold is not None else 0.0), _stream())
    return out, mask


def mask_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """CalcIoU (train_joint_v2.py:683-694) over uint8 masks [N, ...] -> fp64 [N]."""
    _chk(a, "a", torch.uint8); _chk(b, "b", torch.uint8)
    N = a.shape[0]
    per = a.numel() // N
    counts = torch.empty((N, 2), dtype=torch.int64, device=a.device)
    _lib.call("ullsam_mask_iou_counts", a.data_ptr(), b.data_ptr(), counts.data_ptr(), N, per, _stream())
    c = counts.double()
    return (c[:, 0] + 1e-7) / (c[:, 1] + 1e-7)
