"""Training: the differentiable part of one step of the reference's trainer (train_joint_v2.py:943-1100).

    loss, bce, dice = train_step_loss(model, pixel_values, input_ids, attention_mask, (coords, labels), gt_masks)
    loss.backward()        # .grad of vision_model / mlp1 / mlp2 / prompt_encoder / mask_decoder parameters (the LLM is frozen)

is the drop-in for

    outputs = model(pixel_values=..., input_ids=..., attention_mask=..., output_hidden_states=True)         # :990-1010
    image_embeddings = model.vision_model(pixel_values)                                                      # :1020
    sparse, dense = model.prompt_encoder(points, None, None, outputs.hidden_states.repeat(bs, 1, 1, 1))     # :1055-1060
    low, _ = model.mask_decoder(image_embeddings, model.prompt_encoder.get_dense_pe(), sparse, dense, multimask_output=False)
    loss, bce, dice, _ = calc_instance_loss(F.interpolate(low, (S, S), mode="bilinear", align_corners=False), gt, BCELoss(), DiceLoss())

built from three pieces that can be used on their own: `vision_feature_rows` (the vision model), `llm_image_hidden` (pixel_shuffle, mlp1, the
frozen LLM) and `segmentation_loss` (mlp2, prompt encoder, mask decoder, upsample, BCE + Dice).  Supported `trainable_modules`
(train_joint_v2.py:1280-1359): "vision_model", "mlp1", "mlp2", "prompt_encoder", "mask_decoder" -- everything the reference trains.  The
arithmetic is fp32; the model may be fp32 or bf16 (bf16 parameters are widened on use and receive bf16 gradients).  One image per step.

Every arithmetic step, forward and backward, is a HIP kernel (csrc/train.hip for the backward kernels, the generic fp32 matmul and the
training attention; the inference kernels for norms, gathers, sparse embeddings, upsample).  torch supplies the autograd tape and data
movement (reshape / permute / pad / cat / index copies), nothing else.  These are correctness-first kernels: the step is gated on gradients
equal to the reference's autograd (tests/test_train_gpu.py; fixtures tests/golden/train_*.npz made by the reference itself), not on speed.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import _lib, ops

F32 = torch.float32


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _c(t: torch.Tensor) -> torch.Tensor:
    """fp32, contiguous.  bf16 parameters / inputs (the trainer's default `--dtype bfloat16` keeps the model in bf16, train_joint_v2.py:1676) are
    widened here: the arithmetic of the step is fp32 throughout, and autograd hands every bf16 parameter its gradient rounded to bf16."""
    if t.dtype == torch.bfloat16:
        if CACHE_WIDENED and isinstance(t, torch.nn.Parameter) and t.numel() <= (1 << 22):
            return _widened(t)
        t = t.float()
    elif t.dtype != F32:
        raise TypeError(f"the training step computes in fp32 from fp32 or bf16 tensors (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


CACHE_WIDENED = True   # the fp32 copies of a bf16 model's SMALL parameters (norm weights, biases, decoder linears: <= 4 M elements) are kept per parameter object and
#                        re-made when its `_version` or storage changes: a step made ~1200 such copies, 4 us each, all launch-bound
_WIDE = None


def _widened(p: torch.Tensor) -> torch.Tensor:
    global _WIDE
    if _WIDE is None:
        from torch.utils.weak import WeakIdKeyDictionary   # keyed on identity, dies with the parameter (as _TransposeCache)
        _WIDE = WeakIdKeyDictionary()
    e = _WIDE.get(p)
    if e is not None and e[1] == p._version and e[2] == p.data_ptr() and e[0].shape == p.shape:
        return e[0]
    f = p.detach().float().contiguous()
    _WIDE[p] = (f, p._version, p.data_ptr())
    return f


def _mm(A, B, C, M, N, K, sa, sb, sc, batch=1, accumulate=False, bf16=False):
    """C[b](m,n) (+)= sum_k A[b](m,k) B[b](k,n); sa = (batch, m, k) element strides of A, sb = (batch, k, n), sc = (batch, m, n).
    bf16: operands rounded to bf16 on load, bf16 MFMA with fp32 accumulation (autocast's matmul; the attention products of a bf16 model)."""
    if bf16:
        _lib.call("ullsam_train_matmul_bf16", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, batch, *sa, *sb, *sc, int(accumulate), _s())
        return
    tile = 128 if (M >= 64 and N >= 48) else 64
    tiles = -(-M // tile) * -(-N // tile) * batch
    if K >= 512 and tiles <= 128:           # a few output tiles under a long sum: cut k over workgroups (partials added in order: deterministic)
        ks = max(2, min(K // 128, 1024 // tiles, 65535 // batch))      # (round 6: pieces of >= 128 instead of >= 512 terms -- the rel-pos table gradient, 14 x 80 outputs over 5600 terms x 14, 129 -> ~45 us)
        part = torch.empty((ks * batch * M * N,), dtype=F32, device=C.device)
        _lib.call("ullsam_train_matmul_splitk", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, batch, *sa, *sb, *sc, int(accumulate), ks, part.data_ptr(), _s())
        return
    _lib.call("ullsam_train_matmul", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, batch, *sa, *sb, *sc, int(accumulate), _s())


def _mmh(A, B, C, M, N, K, outer, heads, sa, sb, sc, accumulate=False, bf16=False, tri=0):
    """The product over (outer, head) pairs on operands that sit inside [rows, heads x hd] activations (ullsam_train_matmul_heads): entry (o, h) of A starts at
    o sa[0] + (h // sa[2]) sa[1]; sa = (outer, head, head divisor, m, k) element strides, sb = (outer, head, head divisor, k, n), sc = (outer, head, m, n)."""
    _lib.call("ullsam_train_matmul_heads", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, outer, heads, sa[0], sa[1], sa[2], sa[3], sa[4],
              sb[0], sb[1], sb[2], sb[3], sb[4], sc[0], sc[1], sc[2], sc[3], int(accumulate), int(bf16), int(tri), _s())


TRI_CAUSAL = True   # a causal square attention (the LLM's): the matrix-form products skip the 128 x 128 tiles wholly behind the diagonal and clip their sums to the visible part
#                     (ullsam_train_matmul_heads `tri`); the row pass writes zeros there without reading.  False: whole matrices (tests / A-B: same bits)


def _tri(dims) -> bool:
    B, H, KVH, hd, Sq, Sk, causal, kw = dims
    return bool(TRI_CAUSAL and causal == 0 and Sq == Sk)


def _colsum(x2d: torch.Tensor) -> torch.Tensor:
    rows, cols = x2d.shape
    out = torch.zeros((cols,), dtype=F32, device=x2d.device)
    nb = _row_blocks(rows)
    part = torch.empty((nb * cols,), dtype=F32, device=x2d.device) if nb > 1 else None     # per-row-block partials, added in order (no atomics)
    _lib.call("ullsam_train_colsum", x2d.data_ptr(), out.data_ptr(), rows, cols, cols, ops._p(part), _s())
    return out


def _row_blocks(rows: int) -> int:
    """Row blocks of the ordered two-stage column sums (csrc/train.hip colsum_blocks)."""
    return min(64, -(-rows // 64))


def _t2d(x: torch.Tensor) -> torch.Tensor:
    """[R, C] -> [C, R] copy (HIP transpose kernel)."""
    R, C = x.shape
    return ops.transpose(x.reshape(1, R, C), 1, R, C).reshape(C, R)


RECOMPUTE_P = True   # matrix-form attention keeps q, k, v and rebuilds the probability matrix in the backward (one more product and the softmax inside the
#                      backward's row pass) instead of holding it from the forward: -10.8 GB on a ViT-H + 7B-shaped step for ~4 % of its time
MATRIX_ATTN_FROM = 0         # attention through materialised score matrices from Sq * Sk >= this; below it one workgroup per query, whose backward adds dk / dv by
#                              atomics (order-dependent sums): 0 keeps every attention on the matrix form, so that two runs of a step are bit-equal
FUSED_GLOBAL_FWD = True   # bf16 models: the forward value of the ViT's attention blocks (global and windowed) comes from the inference path's kernels on the packed qkv; the backward stays the matrix form
FUSED_CAUSAL_FWD = True   # bf16 models: the frozen LLM's attention FORWARD runs on the inference path's causal kernel (no score matrix); the backward stays the matrix form
FUSED_CAST_TRANSPOSE = True   # LinearBf16Fn: x -> (bf16 x, x^T) and dY -> (bf16 dY, dY^T, column sums) each in one pass (ops.cast_transpose_bf16); False: separate cast / transpose / column-sum launches
INPLACE_ATTN = True  # the ViT / LLM attention products read q / k / v / dO and write out / dq / dk / dv inside the row tensors (ullsam_train_matmul_heads); False: head-major copies around plain batched products (tests / A-B)
MFMA_LINEAR = True   # nn.Linear forward / backward on the fp32 MFMA GEMM of the inference path where its shapes allow (inner dimension % 32 == 0);
#                     the one-output-per-thread matmul of csrc/train.hip otherwise (and always with MFMA_LINEAR = False: tests compare the two)


class LinearFn(Function):
    """y = x W^T + b on rows (nn.Linear); dX = dY W, dW = dY^T X, db = column sums of dY.  Each product is ullsam_gemm's A W^T form: forward
    with W as stored, dX with a transposed copy of W, dW with transposed copies of dY and X."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _c(x), _c(w)
        M, K = x.shape
        N = w.shape[0]
        if MFMA_LINEAR and K % 32 == 0 and N % 4 == 0:
            y = ops.gemm(x, w, None if b is None else _c(b), out_f32=True)
        else:
            y = torch.empty((M, N), dtype=F32, device=x.device)
            _mm(x, w, y, M, N, K, (0, K, 1), (0, 1, K), (0, N, 1))
            if b is not None:
                y = ops.add_cast(y, _c(b).reshape(1, N), F32)
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        M, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if MFMA_LINEAR and N % 32 == 0 and K % 4 == 0:
                dx = ops.gemm(dy, _t2d(w), out_f32=True)                       # [M, N] x ([K, N])^T
            else:
                dx = torch.empty((M, K), dtype=F32, device=x.device)
                _mm(dy, w, dx, M, K, N, (0, N, 1), (0, K, 1), (0, K, 1))
        if ctx.needs_input_grad[1]:
            if MFMA_LINEAR and M % 32 == 0 and K % 4 == 0:
                dw = ops.gemm(_t2d(dy), _t2d(x), out_f32=True)                 # [N, M] x ([K, M])^T
            else:
                dw = torch.empty((N, K), dtype=F32, device=x.device)
                _mm(dy, x, dw, N, K, M, (0, 1, N), (0, K, 1), (0, K, 1))
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = _colsum(dy)
        return dx, dw, db


class _TransposeCache:
    """W^T copies of frozen weights, keyed on the parameter OBJECT through weak references: an entry dies with its weight (a model that is
    freed and rebuilt at the same addresses cannot meet a stale copy), it remembers the weight's `_version` and `data_ptr()` (an in-place
    update or a `.data` swap drops it), and the cache is bounded in BYTES -- a 7B bf16 LLM's transposes are a second copy of its linears
    (~13 GB), which is the budget; `clear()` (also called by `invalidate_transposed_weights`, e.g. after load_state_dict / a LoRA merge done
    through `.data` in place) frees them."""

    def __init__(self, max_bytes: int = 0):
        from torch.utils.weak import WeakIdKeyDictionary   # keyed on identity: Tensor.__eq__ is elementwise
        self._d = WeakIdKeyDictionary()
        self.max_bytes = max_bytes
        self.bytes = 0

    def get(self, w: torch.Tensor) -> torch.Tensor:
        if self.max_bytes < 0:
            self.max_bytes = _default_transpose_budget(w.device) if w.is_cuda else 0
        e = self._d.get(w)
        if e is not None and e[1] == w._version and e[2] == w.data_ptr() and e[0].shape == (w.shape[1], w.shape[0]) and e[0].dtype == w.dtype:
            return e[0]
        if e is not None:
            self.bytes -= e[0].numel() * e[0].element_size()
        t = ops.transpose_to_bf16(w.detach()) if w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 2 and w.is_contiguous() else w.detach().t().contiguous()
        nb = t.numel() * t.element_size()
        if nb > self.max_bytes:                 # (budget 0: nothing is kept, every backward transposes its weight again)
            if e is not None:
                del self._d[w]
            return t
        if self.bytes + nb > self.max_bytes:
            self.clear()
        self._d[w] = (t, w._version, w.data_ptr())
        self.bytes += nb
        return t

    def clear(self):
        self._d.clear()
        self.bytes = 0

    def __len__(self):
        return len(self._d)


_WT_CACHE = _TransposeCache(-1)    # budget -1 = not decided yet: the first use sets min(16 GiB, 1/8 of the device's memory) -- on a 288 GB MI355X the 13 GB of a 7B bf16 LLM's W^T copies
#                                    fit, and re-making them in every backward cost ~3 % of a step (round 4 kept 0 when the step still needed 60 GB); set_transpose_cache_bytes(0) switches it off


def _default_transpose_budget(dev) -> int:
    try:
        total = torch.cuda.get_device_properties(dev).total_memory
    except Exception:
        return 0
    return int(min(16 << 30, total // 8))


def set_transpose_cache_bytes(n: int) -> None:
    """Let the frozen linears keep up to n bytes of W^T copies between steps (0: transposed again in every backward)."""
    _WT_CACHE.clear()
    _WT_CACHE.max_bytes = int(n)


def invalidate_transposed_weights():
    """Drop every cached W^T.  Needed only after edits that autograd's version counter cannot see (writes through `.data` that keep the
    storage: `w.data.copy_()`, a LoRA merge, EMA) -- in-place ops on the parameter itself and reallocation are detected."""
    _WT_CACHE.clear()
    if _WIDE is not None:
        _WIDE.clear()


def _transposed_frozen(w: torch.Tensor) -> torch.Tensor:
    """[N, K] frozen weight -> its [K, N] copy, made once per weight version (the dX GEMM of a frozen bf16 Linear reads W^T row-major)."""
    return _WT_CACHE.get(w)


class FrozenLinearBf16Fn(Function):
    """y = x W^T (+ b) for a FROZEN bf16 weight (the LLM of a bf16 model, train_joint_v2.py:1599,1676: autocast semantics): activations are
    rounded to bf16 at the GEMM's door, the products run on the inference path's bf16 MFMA GEMM with fp32 accumulation and fp32 results,
    forward (W as stored) and backward (dX = dY W, with the cached W^T); there is no weight gradient.  The 1081-row products with N = 4096 (w2 forward, dX of
    wqkv / w1 / w3) are 64 ring tiles: they may run as four K ranges side by side (`splitk_ok`; 228 -> 148 us at K = 14336, tools/probes/splitk_ab.py).  Only here:
    a step holds one image, so there is no batch whose images could get other bits than alone."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        ctx.w = w
        return ops.gemm(ops.cast(x, torch.bfloat16), w.detach(), None if b is None else _c(b), out_f32=True, splitk_ok=True)

    @staticmethod
    def backward(ctx, dy):
        return ops.gemm(ops.cast(_c(dy), torch.bfloat16), _transposed_frozen(ctx.w), out_f32=True, splitk_ok=True), None, None


BF16_LINEAR = True   # a bf16 model's large linears (>= 256 rows, both dimensions % 64) run forward, dX and dW on the bf16 MFMA GEMM -- the trainer's own
#                      semantics: train_joint_v2.py:1665,1676 puts the model in bf16 under autocast(bf16).  False: bf16 weights are widened and the
#                      arithmetic stays fp32 (tests compare the two)


class LinearBf16Fn(Function):
    """y = x W^T (+ b) for a TRAINABLE bf16 weight on the inference path's bf16 MFMA GEMM (fp32 accumulation, fp32 results): forward with W as
    stored; dX = dY W through a transposed copy of W; dW = dY^T X as the GEMM [N, M] x ([K, M])^T over bf16 transposes of dY and X (rows padded
    to a multiple of 64 with zeros); db = column sums of the fp32 dY.  Activations and gradients are rounded to bf16 at each GEMM's door,
    which is what the reference's trainer computes (bf16 model under autocast)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        ctx.has_b = b is not None
        ctx.fused = FUSED_CAST_TRANSPOSE and x.shape[1] % 4 == 0 and w.shape[0] % 4 == 0
        if ctx.fused and ctx.needs_input_grad[1]:      # x leaves its one pass as the GEMM's bf16 operand AND as the x^T the dW product will want (kept instead of x)
            xb, xt, _ = ops.cast_transpose_bf16(x, 64)
            ctx.save_for_backward(xt, w)
        else:
            ctx.fused = False
            xb = ops.cast(x, torch.bfloat16)
            ctx.save_for_backward(xb, w)
        return ops.gemm(xb, w.detach(), None if b is None else _c(b), out_f32=True)

    @staticmethod
    def backward(ctx, dy):
        xs, w = ctx.saved_tensors
        dy = _c(dy)
        dx = dw = db = None
        if ctx.fused:              # one pass over dY: its bf16 copy (dX), its transpose (dW) and its column sums (db)
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            dyb, dyt, db = ops.cast_transpose_bf16(dy, 64, row_major=ctx.needs_input_grad[0], colsum=want_b)
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(dyb, ops.transpose_to_bf16(w.detach()), out_f32=True)
            dw = ops.gemm(dyt, xs, out_f32=True)
            return dx, dw, db
        dyb = ops.cast(dy, torch.bfloat16)
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dyb, ops.transpose_to_bf16(w.detach()), out_f32=True)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm(ops.transpose_to_bf16(dyb, 64), ops.transpose_to_bf16(xs, 64), out_f32=True)   # (from the bf16 copy: the same values, half the bytes read)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = _colsum(dy)
        return dx, dw, db


def _apply_linear(x, w, b):
    """nn.Linear on rows: LinearBf16Fn for the large linears of a bf16 model, LinearFn (fp32 arithmetic) otherwise."""
    if BF16_LINEAR and w.dtype == torch.bfloat16 and x.shape[0] >= 256 and w.shape[1] % 64 == 0 and w.shape[0] % 64 == 0:
        return LinearBf16Fn.apply(x, w, b)
    return LinearFn.apply(x, w, b)


def _frozen_linear(x, w, b):
    """nn.Linear with a frozen weight inside llm_image_hidden: bf16 weights take the bf16 GEMM, fp32 weights the fp32 path."""
    if w.dtype == torch.bfloat16 and not w.requires_grad and w.shape[1] % 64 == 0 and w.shape[0] % 64 == 0:
        return FrozenLinearBf16Fn.apply(x, w, b)
    return _apply_linear(x, w, b)


class LayerNormFn(Function):
    """Row LayerNorm (nn.LayerNorm / LayerNorm2d on NHWC rows / F.layer_norm without affine)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = _c(x)
        y = ops.norm(x, None if w is None else _c(w), None if b is None else _c(b), eps, F32)
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        ctx.affine = w is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        rows, D = x.shape
        dx = torch.empty_like(x)
        dw = torch.zeros((D,), dtype=F32, device=x.device) if ctx.affine else None
        db = torch.zeros((D,), dtype=F32, device=x.device) if ctx.affine else None
        wf = None if w is None else _c(w)   # (a bf16 weight is widened into a temporary: it must outlive the launch)
        ws = torch.empty((2 * rows + 2 * _row_blocks(rows) * D,), dtype=F32, device=x.device) if ctx.affine else None
        _lib.call("ullsam_train_ln_bwd", x.data_ptr(), None if wf is None else wf.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                  ops._p(dw), ops._p(db), rows, D, float(ctx.eps), ops._p(ws), _s())
        del wf
        return dx, dw, db, None


class ActFn(Function):
    """kind 1 = exact GELU (nn.GELU()), 2 = ReLU."""

    @staticmethod
    def forward(ctx, x, kind):
        x = _c(x)
        y = torch.empty_like(x)
        _lib.call("ullsam_train_act", x.data_ptr(), None, y.data_ptr(), x.numel(), kind, _s())
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        _lib.call("ullsam_train_act", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ctx.kind, _s())
        return dx, None


class AddFn(Function):
    """a + b with b's rows broadcast modularly over a's ([rows_a, C] + [rows_b, C], rows_a % rows_b == 0); the gradient of a broadcast
    operand is the sum over its repeats."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.shapes = (a.shape, b.shape)
        C = a.shape[-1]
        return ops.add_cast(a.reshape(-1, C), b.reshape(-1, C), F32).reshape(a.shape)

    @staticmethod
    def backward(ctx, dy):
        sa, sb = ctx.shapes
        dy = _c(dy)
        da = dy if ctx.needs_input_grad[0] else None
        db = None
        if ctx.needs_input_grad[1]:
            nb = math.prod(sb)
            db = dy.reshape(sb) if nb == dy.numel() else _colsum(dy.reshape(-1, nb)).reshape(sb)
        return da, db


class BroadcastRowsFn(Function):
    """[T, C] parameter rows repeated for P prompts (mask_decoder.py:119-120 output_tokens.expand); gradient = sum over the prompts."""

    @staticmethod
    def forward(ctx, x, P):
        ctx.shape = x.shape
        return _c(x).unsqueeze(0).expand(P, *x.shape).contiguous()

    @staticmethod
    def backward(ctx, dy):
        return _colsum(_c(dy).reshape(dy.shape[0], -1)).reshape(ctx.shape), None


class ScaleShiftFn(Function):
    """x * llm_scale_factor + llm_bias (prompt_encoder.py:148), both one-element parameters."""

    @staticmethod
    def forward(ctx, x, s, t):
        x, s, t = _c(x), _c(s), _c(t)
        y = torch.empty_like(x)
        _lib.call("ullsam_train_scale_shift", x.data_ptr(), s.data_ptr(), t.data_ptr(), None, y.data_ptr(), None, None, x.numel(), None, _s())
        ctx.save_for_backward(x, s, t)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s, t = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        ds = torch.zeros_like(s)
        dt = torch.zeros_like(t)
        part = torch.empty((2 * -(-x.numel() // 1024),), dtype=F32, device=x.device)
        _lib.call("ullsam_train_scale_shift", x.data_ptr(), s.data_ptr(), t.data_ptr(), dy.data_ptr(), dx.data_ptr(), ds.data_ptr(),
                  dt.data_ptr(), x.numel(), part.data_ptr(), _s())
        return dx, ds, dt


class AttentionFn(Function):
    """softmax(q k^T / sqrt(hd) + bias + mask) v per head: q [B*Sq, H*hd] rows, k / v [B*Sk, KVH*hd] rows.  The decoder's attention
    (transformer.py:220-242: KVH = H, no mask), InternLM2's (modeling_internlm2.py:383-419: grouped KV heads, causal, padding mask) and the
    ViT's (image_encoder.py:224-240: bias_h [B, H, Sq, Sk/kw] + bias_w [B, H, Sq, kw], the decomposed relative-position terms)."""

    @staticmethod
    def forward(ctx, q, k, v, B, H, KVH, Sq, Sk, causal, key_mask, bias_h, bias_w, kw, bf16=False, fused_out=None):
        q, k, v = _c(q), _c(k), _c(v)
        ctx.bf16 = bool(bf16)     # matrix form only: the score / probability products as autocast(bfloat16) computes them (bf16 models)
        if bias_h is not None:
            bias_h, bias_w = _c(bias_h), _c(bias_w)
        hd = q.shape[-1] // H
        ctx.dims = (B, H, KVH, hd, Sq, Sk, causal, kw)
        if Sq * Sk < MATRIX_ATTN_FROM:       # small: one workgroup per query; the backward recomputes P and adds dk / dv by atomics
            out = torch.empty_like(q)
            ctx.save_for_backward(q, k, v, key_mask, bias_h, bias_w)
            ctx.matrix = False
            ctx.inplace = False
            AttentionFn._launch(q, k, v, None, out, None, None, None, ctx.dims, key_mask, bias_h, bias_w, None, None)
            return out
        ctx.matrix = True
        ctx.recompute = RECOMPUTE_P
        ctx.inplace = INPLACE_ATTN and Sq >= 64 and Sk >= 64 and hd >= 16
        if fused_out is not None and ctx.inplace and RECOMPUTE_P:
            # the caller has the forward value from an inference kernel (the ViT's global blocks: ullsam_vit_attention on the packed qkv); the backward rebuilds P from q / k / v
            ctx.save_for_backward(AttentionFn._scaled(q, 1.0 / math.sqrt(hd)), k, v, key_mask, bias_h, bias_w)
            return _c(fused_out)
        if (FUSED_CAUSAL_FWD and ctx.inplace and ctx.bf16 and RECOMPUTE_P and causal == 0 and bias_h is None and hd == 128 and Sq == Sk and H % KVH == 0):
            # The LLM's forward on the inference path's causal kernel (csrc/attention.hip causal128_attn_kernel: bf16 q / k / v, fp32 online softmax, bf16 probabilities and output --
            # autocast's attention; the reference's additive causal + padding masks): no score matrix in the forward at all.  The backward rebuilds P in matrix form from the
            # saved q scale / k / v as before (RECOMPUTE_P), so it needs nothing from this launch.
            qs = AttentionFn._scaled(q, 1.0 / math.sqrt(hd))
            bf = torch.bfloat16
            kc = k.reshape(B, Sk, KVH, hd).to(bf).permute(0, 2, 1, 3).contiguous()
            vc = v.reshape(B, Sk, KVH, hd).to(bf).permute(0, 2, 1, 3).contiguous()
            out = ops.causal_attention(ops.cast(q, bf), kc, vc, key_mask, B, H, KVH, hd, Sq, Sk, 0).float()
            ctx.save_for_backward(qs, k, v, key_mask, bias_h, bias_w)
            return out
        if ctx.inplace:
            # Matrix form on the activations where they are (ViT, LLM): the products index (image / window, head) pairs inside the [rows, heads x hd] tensors
            # (ullsam_train_matmul_heads; grouped KV heads by h // G) -- no head-major copies of q / k / v, no repeat_kv copies, out written as rows.
            G = H // KVH
            qs = AttentionFn._scaled(q, 1.0 / math.sqrt(hd))
            P = torch.empty((B * H, Sq, Sk), dtype=F32, device=q.device)
            AttentionFn._scores(qs, k, P, ctx.dims, ctx.bf16)
            _lib.call("ullsam_train_attn_rows", P.data_ptr(), None, ops._p(bias_h), ops._p(bias_w), None, None, ops._p(key_mask), B, H, Sq, Sk, kw,
                      causal, 0, _s())
            out = torch.empty_like(q)
            _mmh(P, v, out, Sq, hd, Sk, B, H, (H * Sq * Sk, Sq * Sk, 1, Sk, 1), (Sk * KVH * hd, hd, G, KVH * hd, 1), (Sq * H * hd, hd, H * hd, 1), bf16=ctx.bf16,
                 tri=3 if _tri(ctx.dims) else 0)   # out = P v
            ctx.save_for_backward(qs, k, v, key_mask if RECOMPUTE_P else P, bias_h, bias_w)
            return out
        # Matrix form (csrc/train.hip attn_rows_kernel) for the decoder's few-token attentions: head-major copies, batched matmuls around one row pass; P is kept for the backward.
        qs, kh, vh = AttentionFn._head_major(q, k, v, ctx.dims)
        BH = B * H
        P = torch.empty((BH, Sq, Sk), dtype=F32, device=q.device)
        _mm(qs, kh, P, Sq, Sk, hd, (Sq * hd, hd, 1), (Sk * hd, 1, hd), (Sq * Sk, Sk, 1), batch=BH, bf16=ctx.bf16)          # S = (q scale) k^T
        _lib.call("ullsam_train_attn_rows", P.data_ptr(), None, ops._p(bias_h), ops._p(bias_w), None, None, ops._p(key_mask), B, H, Sq, Sk, kw,
                  causal, 0, _s())                                                                           # S <- P = softmax(S + bias + masks)
        oh = torch.empty_like(qs)
        _mm(P, vh, oh, Sq, hd, Sk, (Sq * Sk, Sk, 1), (Sk * hd, hd, 1), (Sq * hd, hd, 1), batch=BH, bf16=ctx.bf16)          # out = P v
        if RECOMPUTE_P:       # keep q, k, v only; the backward rebuilds P with the same two launches (bit-equal: same kernels, same inputs)
            ctx.save_for_backward(qs, kh, vh, key_mask, bias_h, bias_w)
        else:
            ctx.save_for_backward(qs, kh, vh, P, bias_h, bias_w)
        return oh.permute(0, 2, 1, 3).reshape(B * Sq, H * hd).contiguous()

    @staticmethod
    def _scores(qs, k, P, dims, bf16):
        """P[(b, h)] = qs_h k_{h // G}^T on the row tensors (qs [B*Sq, H*hd], k [B*Sk, KVH*hd])."""
        B, H, KVH, hd, Sq, Sk, _, _ = dims
        _mmh(qs, k, P, Sq, Sk, hd, B, H, (Sq * H * hd, hd, 1, H * hd, 1), (Sk * KVH * hd, hd, H // KVH, 1, KVH * hd), (H * Sq * Sk, Sq * Sk, Sk, 1), bf16=bf16,
             tri=1 if _tri(dims) else 0)

    @staticmethod
    def _head_major(q, k, v, dims):
        """[B*S, Hx*hd] rows -> [B, H, S, hd] copies (data movement); q comes back scaled by 1/sqrt(hd) (a HIP kernel), k / v repeated over
        the KV group (repeat_kv, modeling_internlm2.py:250-259)."""
        B, H, KVH, hd, Sq, Sk, _, _ = dims
        G = H // KVH
        hm = lambda t, S_, Hx: t.reshape(B, S_, Hx, hd).permute(0, 2, 1, 3).contiguous()
        qh, kh, vh = hm(q, Sq, H), hm(k, Sk, KVH), hm(v, Sk, KVH)
        if G > 1:
            kh = kh.unsqueeze(2).expand(B, KVH, G, Sk, hd).reshape(B, H, Sk, hd).contiguous()
            vh = vh.unsqueeze(2).expand(B, KVH, G, Sk, hd).reshape(B, H, Sk, hd).contiguous()
        return AttentionFn._scaled(qh, 1.0 / math.sqrt(hd)), kh, vh

    @staticmethod
    def _scaled(t, factor):
        sc = torch.full((1,), factor, dtype=F32, device=t.device)
        zero = torch.zeros((1,), dtype=F32, device=t.device)
        out = torch.empty_like(t)
        _lib.call("ullsam_train_scale_shift", t.data_ptr(), sc.data_ptr(), zero.data_ptr(), None, out.data_ptr(), None, None, t.numel(), None, _s())
        return out

    @staticmethod
    def _launch(q, k, v, dout, out, dq, dk, dv, dims, key_mask, bias_h, bias_w, dbh, dbw):
        B, H, KVH, hd, Sq, Sk, causal, kw = dims
        sq, sk = (Sq * H * hd, H * hd, hd), (Sk * KVH * hd, KVH * hd, hd)
        _lib.call("ullsam_train_attention", q.data_ptr(), k.data_ptr(), v.data_ptr(), ops._p(dout), ops._p(out), ops._p(dq), ops._p(dk),
                  ops._p(dv), B, H, H // KVH, hd, Sq, Sk, causal, ops._p(key_mask), *sq, *sk, *sk, *sq, 1.0 / math.sqrt(hd),
                  ops._p(bias_h), ops._p(bias_w), ops._p(dbh), ops._p(dbw), kw, _s())

    @staticmethod
    def backward(ctx, dout):
        B, H, KVH, hd, Sq, Sk, causal, kw = ctx.dims
        dout = _c(dout)
        nones = (None,) * 7
        if not ctx.matrix:
            q, k, v, key_mask, bias_h, bias_w = ctx.saved_tensors
            dbh = torch.empty_like(bias_h) if bias_h is not None else None
            dbw = torch.empty_like(bias_w) if bias_w is not None else None
            dq, dk, dv = torch.empty_like(q), torch.zeros_like(k), torch.zeros_like(v)
            AttentionFn._launch(q, k, v, dout, None, dq, dk, dv, ctx.dims, key_mask, bias_h, bias_w, dbh, dbw)
            return (dq, dk, dv) + nones + (dbh, dbw, None, None, None)
        if ctx.inplace:
            return AttentionFn._backward_inplace(ctx, dout)
        qs, kh, vh, P, bias_h, bias_w = ctx.saved_tensors
        G, BH = H // KVH, B * H
        key_mask = None
        if ctx.recompute:     # the fourth saved tensor is the key mask: S = (q scale) k^T again, softmax inside the row pass below (have_p 0)
            key_mask, P = P, torch.empty((BH, Sq, Sk), dtype=F32, device=qs.device)
            _mm(qs, kh, P, Sq, Sk, hd, (Sq * hd, hd, 1), (Sk * hd, 1, hd), (Sq * Sk, Sk, 1), batch=BH, bf16=ctx.bf16)
        dbh = torch.empty_like(bias_h) if bias_h is not None else None
        dbw = torch.empty_like(bias_w) if bias_w is not None else None
        doh = dout.reshape(B, Sq, H, hd).permute(0, 2, 1, 3).contiguous()
        dP = torch.empty_like(P)
        _mm(doh, vh, dP, Sq, Sk, hd, (Sq * hd, hd, 1), (Sk * hd, 1, hd), (Sq * Sk, Sk, 1), batch=BH, bf16=ctx.bf16)        # dP = dO v^T
        _lib.call("ullsam_train_attn_rows", P.data_ptr(), dP.data_ptr(), ops._p(bias_h), ops._p(bias_w), ops._p(dbh), ops._p(dbw), ops._p(key_mask), B, H,
                  Sq, Sk, kw, causal, 0 if ctx.recompute else 1, _s())                                       # dP <- dS = P (dP - sum_j P_j dP_j)
        dvh, dkh, dqs = torch.empty_like(kh), torch.empty_like(kh), torch.empty_like(qs)
        _mm(P, doh, dvh, Sk, hd, Sq, (Sq * Sk, 1, Sk), (Sq * hd, hd, 1), (Sk * hd, hd, 1), batch=BH, bf16=ctx.bf16)        # dV = P^T dO
        _mm(dP, qs, dkh, Sk, hd, Sq, (Sq * Sk, 1, Sk), (Sq * hd, hd, 1), (Sk * hd, hd, 1), batch=BH, bf16=ctx.bf16)        # dK = dS^T (q scale)
        _mm(dP, kh, dqs, Sq, hd, Sk, (Sq * Sk, Sk, 1), (Sk * hd, hd, 1), (Sq * hd, hd, 1), batch=BH, bf16=ctx.bf16)        # d(q scale) = dS k
        dqh = AttentionFn._scaled(dqs, 1.0 / math.sqrt(hd))
        if G > 1:                                                                                   # the gradient of repeat_kv: sum over the group
            red = lambda t: _colsum(t.reshape(B, KVH, G, Sk * hd).permute(2, 0, 1, 3).reshape(G, -1).contiguous()).reshape(B, KVH, Sk, hd)
            dkh, dvh = red(dkh), red(dvh)
        back = lambda t, S_, Hx: t.permute(0, 2, 1, 3).reshape(B * S_, Hx * hd).contiguous()
        return (back(dqh, Sq, H), back(dkh, Sk, KVH), back(dvh, Sk, KVH)) + nones + (dbh, dbw, None, None, None)


def _attn_backward_inplace(ctx, dout):
    """The backward of the in-place matrix form: dP = dO v^T, one row pass (dS, bias-gradient rows), dV = P^T dO, dK = dS^T (q scale), d(q scale) = dS k, every product on
    the row tensors; with grouped KV heads dK / dV are formed per query head and summed over the group in order (repeat_kv's gradient)."""
    B, H, KVH, hd, Sq, Sk, causal, kw = ctx.dims
    qs, k, v, P, bias_h, bias_w = ctx.saved_tensors
    G = H // KVH
    key_mask = None
    if ctx.recompute:
        key_mask, P = P, torch.empty((B * H, Sq, Sk), dtype=F32, device=qs.device)
        AttentionFn._scores(qs, k, P, ctx.dims, ctx.bf16)
    dbh = torch.empty_like(bias_h) if bias_h is not None else None
    dbw = torch.empty_like(bias_w) if bias_w is not None else None
    sP, sPT = (H * Sq * Sk, Sq * Sk, 1, Sk, 1), (H * Sq * Sk, Sq * Sk, 1, 1, Sk)        # P / dS as [m = q][k = key] and transposed [m = key][k = q]
    rows_q = (Sq * H * hd, hd, 1, H * hd, 1)                                               # dO / qs as the B operand [k = q][n = d] resp. the A operand [m = q][k = d]
    dP = torch.empty_like(P)
    tri = _tri(ctx.dims)
    _mmh(dout, v, dP, Sq, Sk, hd, B, H, rows_q, (Sk * KVH * hd, hd, G, 1, KVH * hd), (H * Sq * Sk, Sq * Sk, Sk, 1), bf16=ctx.bf16, tri=1 if tri else 0)   # dP = dO v^T
    _lib.call("ullsam_train_attn_rows", P.data_ptr(), dP.data_ptr(), ops._p(bias_h), ops._p(bias_w), ops._p(dbh), ops._p(dbw), ops._p(key_mask), B, H,
              Sq, Sk, kw, causal, 0 if ctx.recompute else 1, _s())                                                          # dP <- dS
    if G == 1:
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        kv_c = (Sk * H * hd, hd, H * hd, 1)
    else:
        dk = torch.empty((B, H, Sk, hd), dtype=F32, device=qs.device)
        dv = torch.empty_like(dk)
        kv_c = (H * Sk * hd, Sk * hd, hd, 1)
    _mmh(P, dout, dv, Sk, hd, Sq, B, H, sPT, rows_q, kv_c, bf16=ctx.bf16, tri=2 if tri else 0)                                # dV = P^T dO
    _mmh(dP, qs, dk, Sk, hd, Sq, B, H, sPT, rows_q, kv_c, bf16=ctx.bf16, tri=2 if tri else 0)                                 # dK = dS^T (q scale)
    dqs = torch.empty_like(qs)
    _mmh(dP, k, dqs, Sq, hd, Sk, B, H, sP, (Sk * KVH * hd, hd, G, KVH * hd, 1), (Sq * H * hd, hd, H * hd, 1), bf16=ctx.bf16, tri=3 if tri else 0)    # d(q scale) = dS k
    dq = AttentionFn._scaled(dqs, 1.0 / math.sqrt(hd))
    if G > 1:                                                                               # the gradient of repeat_kv: sum over the group, then back to rows
        red = lambda t: _colsum(t.reshape(B, KVH, G, Sk * hd).permute(2, 0, 1, 3).reshape(G, -1).contiguous()).reshape(B, KVH, Sk, hd)
        back = lambda t: t.permute(0, 2, 1, 3).reshape(B * Sk, KVH * hd).contiguous()
        dk, dv = back(red(dk)), back(red(dv))
    return (dq, dk, dv) + (None,) * 7 + (dbh, dbw, None, None, None)


AttentionFn._backward_inplace = staticmethod(_attn_backward_inplace)


class SplitFn(Function):
    """x [R, n_0 + n_1 + ..., C] -> contiguous pieces [R, n_i * C] along the middle axis (the q / k / v split of a fused qkv projection: image_encoder.py:232-233,
    modeling_internlm2.py:361-370); data movement only.  The backward is ONE concatenation of the incoming gradients -- torch's own select / slice backward builds a
    zero tensor of the whole qkv per piece and adds them up (three zero fills, three copies and two adds of the full tensor per attention)."""

    @staticmethod
    def forward(ctx, x, sizes):
        ctx.sizes = tuple(sizes)
        ctx.shape = x.shape
        R, _, C = x.shape
        outs, o = [], 0
        for n in sizes:
            outs.append(x[:, o:o + n].reshape(R, n * C).contiguous())
            o += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        R, _, C = ctx.shape
        return torch.cat([_c(g).reshape(R, n, C) for g, n in zip(grads, ctx.sizes)], dim=1), None


class GatherRowsFn(Function):
    """table[idx] (get_rel_pos, image_encoder.py:303-322, without interpolation); the gradient adds the rows back."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx.to(torch.int32).contiguous())
        ctx.shape = table.shape
        return _c(table)[idx].contiguous()

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dt = torch.zeros(ctx.shape, dtype=F32, device=dy.device)
        _lib.call("ullsam_train_index_add_rows", _c(dy).data_ptr(), idx.data_ptr(), dt.data_ptr(), idx.numel(), ctx.shape[1], ctx.shape[0], _s())
        return dt, None


class BmmNTFn(Function):
    """C[b] = A[b] @ B[b]^T for A [b, M, K], B [b, N, K] (the einsums of add_decomposed_rel_pos, image_encoder.py:351-353)."""

    @staticmethod
    def forward(ctx, A, Bm):
        A, Bm = _c(A), _c(Bm)
        nb, M, K = A.shape
        N = Bm.shape[1]
        C = torch.empty((nb, M, N), dtype=F32, device=A.device)
        _mm(A, Bm, C, M, N, K, (M * K, K, 1), (N * K, 1, K), (M * N, N, 1), batch=nb)
        ctx.save_for_backward(A, Bm)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, Bm = ctx.saved_tensors
        dC = _c(dC)
        nb, M, K = A.shape
        N = Bm.shape[1]
        dA, dB = torch.empty_like(A), torch.empty_like(Bm)
        _mm(dC, Bm, dA, M, K, N, (M * N, N, 1), (N * K, K, 1), (M * K, K, 1), batch=nb)          # dA = dC B
        _mm(dC, A, dB, N, K, M, (M * N, 1, N), (M * K, K, 1), (N * K, K, 1), batch=nb)           # dB = dC^T A
        return dA, dB


class Im2col3x3Fn(Function):
    """3x3 / pad 1 patches of NHWC rows (the neck's second convolution as im2col + Linear, image_encoder.py:96-102)."""

    @staticmethod
    def forward(ctx, x, B, H, W):
        x = _c(x)
        ctx.dims = (B, H, W, x.shape[-1])
        return ops.im2col3x3(x, B, H, W, x.shape[-1])

    @staticmethod
    def backward(ctx, dcols):
        B, H, W, C = ctx.dims
        dx = torch.empty((B * H * W, C), dtype=F32, device=dcols.device)
        _lib.call("ullsam_train_col2im3x3", _c(dcols).data_ptr(), dx.data_ptr(), B, H, W, C, _s())
        return dx, None, None, None


class RMSNormFn(Function):
    """InternLM2RMSNorm (modeling_internlm2.py:75-89)."""

    @staticmethod
    def forward(ctx, x, w, eps):
        x, w = _c(x), _c(w)
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return ops.norm(x, w, None, eps, F32, rms=True)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        rows, D = x.shape
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w) if ctx.needs_input_grad[1] else None
        _lib.call("ullsam_train_rmsnorm_bwd", x.data_ptr(), w.data_ptr(), _c(dy).data_ptr(), dx.data_ptr(), ops._p(dw), rows, D, float(ctx.eps), _s())
        return dx, dw, None


class RoPEFn(Function):
    """apply_rotary_pos_emb (modeling_internlm2.py:233-247) on rows [tokens, heads*hd]; pos int32 [tokens]."""

    @staticmethod
    def forward(ctx, x, pos, cos, sin, heads):
        x = _c(x)
        ctx.save_for_backward(pos, cos, sin)
        ctx.heads = heads
        return RoPEFn._run(x, pos, cos, sin, heads, 0)

    @staticmethod
    def _run(x, pos, cos, sin, heads, adjoint):
        out = torch.empty_like(x)
        _lib.call("ullsam_train_rope", x.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), out.data_ptr(), x.shape[0], heads,
                  x.shape[1] // heads, cos.shape[0], adjoint, _s())
        return out

    @staticmethod
    def backward(ctx, dy):
        pos, cos, sin = ctx.saved_tensors
        return RoPEFn._run(_c(dy), pos, cos, sin, ctx.heads, 1), None, None, None, None


class SwiGLUFn(Function):
    """silu(g) * u (modeling_internlm2.py:617)."""

    @staticmethod
    def forward(ctx, g, u):
        g, u = _c(g), _c(u)
        ctx.save_for_backward(g, u)
        out = torch.empty_like(g)
        _lib.call("ullsam_train_swiglu", g.data_ptr(), u.data_ptr(), None, out.data_ptr(), None, None, g.numel(), _s())
        return out

    @staticmethod
    def backward(ctx, dy):
        g, u = ctx.saved_tensors
        dg, du = torch.empty_like(g), torch.empty_like(u)
        _lib.call("ullsam_train_swiglu", g.data_ptr(), u.data_ptr(), _c(dy).data_ptr(), None, dg.data_ptr(), du.data_ptr(), g.numel(), _s())
        return dg, du


class HyperMasksFn(Function):
    """masks[p] = hyper[p] [M, c] @ up[p]^T [c, pixels] (mask_decoder.py:146-147), up as rows [P, pixels, c]."""

    @staticmethod
    def forward(ctx, hyper, up):
        hyper, up = _c(hyper), _c(up)
        P, M, c = hyper.shape
        npix = up.shape[1]
        out = torch.empty((P, M, npix), dtype=F32, device=up.device)
        _mm(hyper, up, out, M, npix, c, (M * c, c, 1), (npix * c, 1, c), (M * npix, npix, 1), batch=P)
        ctx.save_for_backward(hyper, up)
        return out

    @staticmethod
    def backward(ctx, dm):
        hyper, up = ctx.saved_tensors
        dm = _c(dm)
        P, M, c = hyper.shape
        npix = up.shape[1]
        dh = torch.empty_like(hyper)
        du = torch.empty_like(up)
        _mm(dm, up, dh, M, c, npix, (M * npix, npix, 1), (npix * c, c, 1), (M * c, c, 1), batch=P)           # dH = dM U
        _mm(dm, hyper, du, npix, c, M, (M * npix, 1, npix), (M * c, c, 1), (npix * c, c, 1), batch=P)        # dU = dM^T H
        return dh, du


class SparseEmbedFn(Function):
    """Point / box prompts -> sparse embeddings (prompt_encoder.py:76-103): random-Fourier PE of the click (of the two corners of a box) + the label's
    embedding row; the padding point (appended when there are no boxes, :84-88,181) and label -1 take not_a_point_embed alone; box corners take
    point_embeddings[2] / [3].  table = [not_a_point, point_embeddings 0..3] (5 rows); the gradient is a row sum into that table (the PE has no parameters)."""

    @staticmethod
    def forward(ctx, table, coords, labels, boxes, G, img_hw):
        table = _c(table)
        C = table.shape[1]
        P = coords.shape[0] if coords is not None else boxes.shape[0]
        Np = coords.shape[1] if coords is not None else 0
        pad = 1 if (coords is not None and boxes is None) else 0
        out = ops.sparse_embed(coords, labels, boxes, G, table, P, Np, pad, C, img_hw[1], img_hw[0])
        dev = table.device
        cols = []
        if coords is not None:
            cols.append(labels.to(torch.int32) + 1)
        if pad:
            cols.append(torch.zeros((P, 1), dtype=torch.int32, device=dev))
        if boxes is not None:
            cols.append(torch.tensor([[3, 4]], dtype=torch.int32, device=dev).expand(P, 2))
        ctx.save_for_backward(torch.cat(cols, 1).contiguous())
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = _c(dy)
        dt = torch.zeros((5, ctx.C), dtype=F32, device=dy.device)
        _lib.call("ullsam_train_index_add_rows", dy.data_ptr(), idx.data_ptr(), dt.data_ptr(), idx.numel(), ctx.C, 5, _s())
        return dt, None, None, None, None, None


class ResizeFn(Function):
    """F.interpolate(x, (S, S), mode="bilinear", align_corners=False) on [P, 1, h, w]."""

    @staticmethod
    def forward(ctx, x, out_hw):
        x = _c(x)
        ctx.in_shape = x.shape
        ctx.out_hw = out_hw
        return ops.resize_bilinear(x, out_hw)[0]

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        shp = ctx.in_shape
        dx = torch.zeros(shp, dtype=F32, device=dy.device)
        planes = dx.numel() // (shp[-1] * shp[-2])
        _lib.call("ullsam_train_resize_bwd", dy.data_ptr(), dx.data_ptr(), planes, shp[-2], shp[-1], ctx.out_hw[0], ctx.out_hw[1], _s())
        return dx, None


class SegLossFn(Function):
    """calc_instance_loss(pred, gt, BCELoss(), DiceLoss()) (train_joint_v2.py:774-812): returns [3] = (total, bce, dice); only the total
    is differentiable."""

    @staticmethod
    def forward(ctx, pred, gt, smooth):
        pred, gt = _c(pred), _c(gt)
        P = pred.shape[0] * pred.shape[1]
        npix = pred.numel() // P
        sums = torch.zeros((P, 4), dtype=F32, device=pred.device)
        losses = torch.empty((3,), dtype=F32, device=pred.device)
        part = torch.empty((P * 4 * -(-npix // 1024),), dtype=F32, device=pred.device)
        _lib.call("ullsam_train_seg_loss", pred.data_ptr(), gt.data_ptr(), sums.data_ptr(), losses.data_ptr(), P, npix, float(smooth), part.data_ptr(), _s())
        ctx.save_for_backward(pred, gt, sums)
        ctx.dims = (P, npix, float(smooth))
        return losses

    @staticmethod
    def backward(ctx, dl):
        pred, gt, sums = ctx.saved_tensors
        P, npix, smooth = ctx.dims
        g = _c(dl)[:1].contiguous()       # the bce / dice entries are reported values (the reference returns them for logging)
        dx = torch.empty_like(pred)
        _lib.call("ullsam_train_seg_loss_bwd", pred.data_ptr(), gt.data_ptr(), sums.data_ptr(), g.data_ptr(), dx.data_ptr(), P, npix, smooth, _s())
        return dx, None, None


class LMLossFn(Function):
    """The language-model loss of InternLM2ForCausalLM.forward (modeling_internlm2.py:1081-1096) on hidden rows: logits = output(h).float() on the inference
    path's GEMM (bf16 weights: bf16 MFMA, fp32 weights: exact-fp32 MFMA), CrossEntropyLoss() over them (labels == -100 ignored, mean over the others; HIP
    kernels, ordered sums).  The [rows, 92553] logits are NOT kept (400 MB per image): the backward rebuilds them, forms (softmax - onehot) / #labelled in a
    zero-padded [rows, V rounded up to 64] buffer and multiplies by the head's weight.  The head is frozen in every setting of the reference's trainer
    (setup_model_params, train_joint_v2.py:1280-1359): it receives no gradient here.  The trainer's segmentation branch adds this loss as `0 * loss`
    (:1096): an incoming gradient of exactly zero returns zeros without the two GEMMs (one scalar read back per step)."""

    @staticmethod
    def _logits(h, w):
        if w.dtype == torch.bfloat16:
            return ops.gemm(ops.cast(h, torch.bfloat16), w.detach(), None, out_f32=True)
        return ops.gemm(h, w.detach(), None, out_f32=True)

    @staticmethod
    def forward(ctx, h, w, labels):
        h = _c(h)
        labels = labels.to(torch.int64).contiguous()
        R, V = h.shape[0], w.shape[0]
        logits = LMLossFn._logits(h, w)
        lse = torch.empty((R,), dtype=F32, device=h.device)
        rows = torch.empty((R,), dtype=F32, device=h.device)
        out2 = torch.empty((2,), dtype=F32, device=h.device)
        _lib.call("ullsam_train_cross_entropy", logits.data_ptr(), V, labels.data_ptr(), lse.data_ptr(), rows.data_ptr(), out2.data_ptr(), R, V, _s())
        ctx.save_for_backward(h, w, labels, lse, out2)
        return out2[0].clone()

    @staticmethod
    def backward(ctx, dl):
        h, w, labels, lse, out2 = ctx.saved_tensors
        if float(dl) == 0.0:                                  # `0 * loss + seg_loss`: the gradient is exactly zero
            return torch.zeros_like(h), None, None
        R, D = h.shape
        V = w.shape[0]
        Vp = -(-V // 64) * 64
        logits = LMLossFn._logits(h, w)
        g = _c(dl).reshape(1).to(F32)
        dlog = torch.empty((R, Vp), dtype=F32, device=h.device)
        _lib.call("ullsam_train_cross_entropy_bwd", logits.data_ptr(), V, labels.data_ptr(), lse.data_ptr(), out2.data_ptr(), g.data_ptr(), dlog.data_ptr(), Vp, R, V, _s())
        del logits
        if w.dtype == torch.bfloat16 and D % 64 == 0:
            dx = ops.gemm(ops.cast(dlog, torch.bfloat16), ops.transpose_to_bf16(w.detach(), 64), out_f32=True)   # [R, Vp] x ([D, Vp])^T; W^T's pad columns are zeros
        else:
            dx = torch.empty((R, D), dtype=F32, device=h.device)
            _mm(dlog, _c(w.detach()).float() if w.dtype != F32 else _c(w.detach()), dx, R, D, V, (0, Vp, 1), (0, D, 1), (0, D, 1))
        return dx, None, None


def lm_loss(lm, hidden_all: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """`.loss` of the reference's forward as a differentiable scalar (see LMLossFn): hidden_all [B, S, D] carries the graph (mlp1 through the frozen LLM);
    shift_logits = logits[..., :-1, :] against labels[..., 1:]: the last position's logits are never formed."""
    B, S, D = hidden_all.shape
    w = lm.output.weight
    if w.requires_grad:
        raise NotImplementedError("a trainable LM head: the reference's trainer freezes the language model in every setting (train_joint_v2.py:1280-1359)")
    h = hidden_all[:, :-1].reshape(B * (S - 1), D)
    lab = labels[:, 1:].reshape(-1)
    if not lab.is_cuda:   # host labels (what the trainer's collate hands over): torch's CrossEntropyLoss raises on a label outside {-100} U [0, V) -- checked here, where it costs no device sync
        bad = (lab != -100) & ((lab < 0) | (lab >= w.shape[0]))
        if bool(bad.any()):
            raise ValueError(f"lm_loss: label {int(lab[bad][0])} outside [0, {w.shape[0]}) (and not the ignore index -100)")
    # (labels already on the GPU are not read back: there a label outside the range is treated as ignored -- include/ullsam_hip.h, ullsam_train_cross_entropy)
    return LMLossFn.apply(h, w, lab.to(h.device))


# ---------------------------------------------------------------------------------------------------------------------------------------
def _linear(lin, x):
    return _apply_linear(x, lin.weight, lin.bias)


def _ln(norm, x):
    return LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)


def _attention(at, q, k, v, B, Sq, Sk):
    """transformer.py:220-242: projections, per-head softmax attention, output projection; rows [B*S, C]."""
    q, k, v = _linear(at.q_proj, q), _linear(at.k_proj, k), _linear(at.v_proj, v)
    return _linear(at.out_proj, AttentionFn.apply(q, k, v, B, at.num_heads, at.num_heads, Sq, Sk, -1, None, None, None, 0))


def _two_way_transformer(tr, keys, key_pe, tokens):
    """TwoWayTransformer.forward (transformer.py:62-108) on rows: keys [P*N, C], key_pe [N, C] (constant), tokens [P*T, C]."""
    P = tokens.shape[0]
    T, N, C = tokens.shape[1], key_pe.shape[0], tokens.shape[2]
    qpe = tokens.reshape(P * T, C)
    queries = qpe
    for blk in tr.layers:                                                        # TwoWayAttentionBlock.forward :151-184
        if blk.skip_first_layer_pe:
            queries = _attention(blk.self_attn, queries, queries, queries, P, T, T)
        else:
            q = AddFn.apply(queries, qpe)
            queries = AddFn.apply(queries, _attention(blk.self_attn, q, q, queries, P, T, T))
        queries = _ln(blk.norm1, queries)
        q = AddFn.apply(queries, qpe)
        k = AddFn.apply(keys, key_pe)
        queries = _ln(blk.norm2, AddFn.apply(queries, _attention(blk.cross_attn_token_to_image, q, k, keys, P, T, N)))
        m = blk.mlp
        queries = _ln(blk.norm3, AddFn.apply(queries, _linear(m.lin2, ActFn.apply(_linear(m.lin1, queries), m.act_code))))
        q = AddFn.apply(queries, qpe)
        k = AddFn.apply(keys, key_pe)
        keys = _ln(blk.norm4, AddFn.apply(keys, _attention(blk.cross_attn_image_to_token, k, q, queries, P, N, T)))
    q = AddFn.apply(queries, qpe)
    k = AddFn.apply(keys, key_pe)
    queries = _ln(tr.norm_final_attn, AddFn.apply(queries, _attention(tr.final_attn_token_to_image, q, k, keys, P, T, N)))
    return queries.reshape(P, T, C), keys


def _conv_transpose_k2s2(ct, x_rows):
    """nn.ConvTranspose2d(kernel 2, stride 2) on NHWC rows: every input pixel makes a 2x2 block of output pixels = one Linear with the
    weight viewed as [(ky, kx, cout), cin]; rows out: [rows_in * 4, cout] in (pixel, ky, kx) order."""
    cin, cout = ct.weight.shape[0], ct.weight.shape[1]
    w = ct.weight.permute(2, 3, 1, 0).reshape(4 * cout, cin)                     # data movement only (autograd routes the gradient back)
    b = BroadcastRowsFn.apply(ct.bias, 4).reshape(4 * cout)
    return _apply_linear(x_rows, w, b).reshape(-1, cout)


def dense_feature_rows(model, hidden: torch.Tensor) -> torch.Tensor:
    """text_aware_dense_feature (modeling_internvl_sam.py:253-270) up to NHWC rows: hidden [B, n_tok, D_llm] -> [B, H*W, 256] (H = W = 64)."""
    B, n, D = hidden.shape
    ln, l1, l3 = model.mlp2[0], model.mlp2[1], model.mlp2[3]
    x = _linear(l3, ActFn.apply(_linear(l1, _ln(ln, hidden.reshape(B * n, D))), 1))
    g = int(math.sqrt(n))
    r = model.downsample_ratio
    f = x.reshape(B, g, g, -1)                                                  # the reference's reshapes / permutes, verbatim data movement
    if model.ps_version != "v1":
        f = f.permute(0, 2, 1, 3).contiguous()
    n_, h, w, c = f.shape
    f = f.reshape(n_, h, int(w / r), int(c * r)).permute(0, 2, 1, 3).contiguous()
    f = f.reshape(n_, int(w / r), int(h / r), int(c * (r * r)))                 # = NHWC of the reference's [B, 256, 64, 64]
    return f.reshape(B, -1, f.shape[-1])


def segmentation_loss(model, llm_hidden: torch.Tensor, image_embeddings: Optional[torch.Tensor], points: Tuple[torch.Tensor, torch.Tensor],
                      gt_masks: torch.Tensor, smooth: float = 1e-7, image_rows: Optional[torch.Tensor] = None
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """llm_hidden fp32 [1, n_img_tokens, D_llm] (the LLM's last hidden state over the image tokens, a constant here); image_embeddings
    fp32 [1, C, H, W] (constant) or, instead, image_rows = vision_feature_rows(...) [1, H*W, C] (differentiable: trains the vision model);
    points = (coords [P, n, 2], labels [P, n]) for P instances; gt_masks fp32 [P, 1, S, S].
    Returns (total, bce, dice) as 0-d tensors; `total.backward()` fills the gradients."""
    pe, md = model.prompt_encoder, model.mask_decoder
    dev = llm_hidden.device
    coords, labels = points[0].to(dev).float().contiguous(), points[1].to(dev).to(torch.int32).contiguous()
    P = coords.shape[0]
    Hh, Ww = pe.image_embedding_size
    N, C = Hh * Ww, pe.embed_dim
    # dense prompt: mlp2 -> inverse pixel shuffle -> per-pixel LayerNorm (no affine) * scale + bias, the same rows for every instance
    rows = dense_feature_rows(model, _c(llm_hidden)).reshape(N, C)
    dense = ScaleShiftFn.apply(LayerNormFn.apply(rows, None, None, 1e-5), pe.llm_scale_factor, pe.llm_bias)
    # sparse prompt
    table = torch.cat([pe.not_a_point_embed.weight] + [e.weight for e in pe.point_embeddings], 0)
    sparse = SparseEmbedFn.apply(table, coords, labels, None, pe.pe_layer.G(), pe.input_image_size)
    # mask decoder (mask_decoder.py:112-149)
    out_tok = torch.cat([md.iou_token.weight, md.mask_tokens.weight], 0)
    tokens = torch.cat([BroadcastRowsFn.apply(out_tok, P), sparse], 1)
    if image_rows is not None:                                                                               # vision_feature_rows(...): differentiable
        img_rows = image_rows.reshape(N, C)
    else:
        img_rows = ops.transpose(_c(image_embeddings).reshape(1, C, N), 1, C, N).reshape(N, C)               # NCHW -> rows (constant)
    src = AddFn.apply(BroadcastRowsFn.apply(dense, P).reshape(P * N, C), img_rows)                          # repeat_interleave(image) + dense
    hs, keys = _two_way_transformer(md.transformer, src, pe.dense_pe_tokens(), tokens)
    up0, ln, up1 = md.output_upscaling[0], md.output_upscaling[1], md.output_upscaling[3]
    c8 = C // 8
    u = _conv_transpose_k2s2(up0, keys)                                                                      # [P*N*4, C/4]
    u = ActFn.apply(LayerNormFn.apply(u, ln.weight, ln.bias, ln.eps), 1)
    u = ActFn.apply(_conv_transpose_k2s2(up1, u), 1)                                                         # [P*N*16, C/8]
    nm = md.num_mask_tokens
    hyper = []
    for i in range(nm):
        x = hs[:, 1 + i, :]
        mlp = md.output_hypernetworks_mlps[i]
        for j, l in enumerate(mlp.layers):
            x = _linear(l, x)
            if j < mlp.num_layers - 1:
                x = ActFn.apply(x, 2)
        hyper.append(x)
    hyper = torch.stack(hyper, 1)                                                                            # [P, 4, C/8]
    m = HyperMasksFn.apply(hyper, u.reshape(P, N * 16, c8))                                                  # [P, 4, (i, j, ky, kx, ky2, kx2)]
    m = m.reshape(P, nm, Hh, Ww, 2, 2, 2, 2).permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(P, nm, 4 * Hh, 4 * Ww)  # y = 4i + 2ky + ky2, x likewise
    low = m[:, 0:1].contiguous()                                                                             # multimask_output=False
    S = model.vision_model.img_size
    pred = ResizeFn.apply(low, (S, S))
    losses = SegLossFn.apply(pred, _c(gt_masks.to(dev).float()), smooth)
    return losses[0], losses[1], losses[2]


# ---------------------------------------------------------------------------------------------------------------------------------------
# Second slice: the projector mlp1 and the (frozen) LLM between the vision features and the hidden states the first slice starts from.
def llm_image_hidden(model, vit_feature_rows: torch.Tensor, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                     return_all: bool = False):
    """The LLM's last hidden state over the image tokens, differentiable with respect to `mlp1` (and to the vision features):
    extract_feature's pixel_shuffle + mlp1 (modeling_internvl_sam.py:226-251), the image-token splice of forward (:136-158), InternLM2's
    layers (modeling_internlm2.py:598-618, 345-419, 75-89; frozen weights receive no gradient work) and the image-token slice of :195-205.
    vit_feature_rows fp32 [B, 64*64, 256] = the vision model's output as NHWC rows (a constant here: the reference computes it under
    no_grad, :243-244); input_ids [B, S] with the <IMG_CONTEXT> run; returns [B, n_img_tokens, D_llm]."""
    lm = model.language_model
    cfg = lm.config
    B, S = input_ids.shape
    dev = vit_feature_rows.device
    g = int(math.sqrt(vit_feature_rows.shape[1]))
    r = model.downsample_ratio
    f = _c(vit_feature_rows).reshape(B, g, g, -1)                               # pixel_shuffle, ps_version v2: data movement, as the reference writes it
    n, h, w, c = f.shape
    f = f.reshape(n, h, int(w * r), int(c / r)).permute(0, 2, 1, 3).contiguous()
    f = f.reshape(n, int(w * r), int(h * r), int(c / (r * r)))
    if model.ps_version != "v1":
        f = f.permute(0, 2, 1, 3).contiguous()
    f = f.reshape(-1, f.shape[-1])
    ln, l1, l3 = model.mlp1[0], model.mlp1[1], model.mlp1[3]
    vit_embeds = _linear(l3, ActFn.apply(_linear(l1, _ln(ln, f)), 1))            # [B*n_img, D]
    D = vit_embeds.shape[-1]
    ids = input_ids.reshape(-1)
    sel = ids == model.img_context_token_id
    n_sel = int(sel.sum())
    if n_sel != vit_embeds.shape[0]:
        raise ValueError(f"{n_sel} <IMG_CONTEXT> tokens for {vit_embeds.shape[0]} image embeddings")
    x = lm.model.tok_embeddings.weight.detach()[ids].float().clone()             # frozen embedding rows (a gather)
    x = x.index_put((sel.nonzero(as_tuple=True)[0],), vit_embeds)                # input_embeds[selected] = vit_embeds  (:150-152)
    H, KVH = cfg.num_attention_heads, cfg.num_key_value_heads
    hd, G = cfg.hidden_size // H, H // KVH
    cos, sin = lm.model.rope_tables(S, dev)
    pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(B).contiguous()  # position_ids default (modeling_internlm2.py:893-898)
    key_mask = None if attention_mask is None else attention_mask.to(torch.int32).contiguous()
    for layer in lm.model.layers:
        at, ff = layer.attention, layer.feed_forward
        xn = RMSNormFn.apply(x, layer.attention_norm.weight, layer.attention_norm.variance_epsilon)
        qkv = _frozen_linear(xn, at.wqkv.weight, at.wqkv.bias).reshape(B * S * KVH, G + 2, hd)   # 'b q (h gs d) -> b q h gs d' (:361-366)
        q, k, v = SplitFn.apply(qkv, (G, 1, 1))                                                   # per (row, kv head): its G query heads, its key, its value
        q = RoPEFn.apply(q.reshape(B * S, H * hd), pos, cos, sin, H)
        k = RoPEFn.apply(k.reshape(B * S, KVH * hd), pos, cos, sin, KVH)
        v = v.reshape(B * S, KVH * hd)
        a = AttentionFn.apply(q, k, v, B, H, KVH, S, S, 0, key_mask, None, None, 0, BF16_LINEAR and at.wqkv.weight.dtype == torch.bfloat16)
        x = AddFn.apply(x, _frozen_linear(a, at.wo.weight, at.wo.bias))
        xn = RMSNormFn.apply(x, layer.ffn_norm.weight, layer.ffn_norm.variance_epsilon)
        hmid = SwiGLUFn.apply(_frozen_linear(xn, ff.w1.weight, None), _frozen_linear(xn, ff.w3.weight, None))
        x = AddFn.apply(x, _frozen_linear(hmid, ff.w2.weight, None))
    x = RMSNormFn.apply(x, lm.model.norm.weight, lm.model.norm.variance_epsilon).reshape(B, S, D)
    idx = sel.reshape(B, S).nonzero(as_tuple=True)[1]
    start, end = int(idx.min()), int(idx.max()) + 1                              # one span for the batch, as the reference takes it (:198-201)
    if return_all:
        return x[:, start:end], x
    return x[:, start:end]


# ---------------------------------------------------------------------------------------------------------------------------------------
# Third slice: the vision model (ImageEncoderViT.forward, image_encoder.py:106-117), reached through the decoder's image embedding.
def vision_feature_rows(enc, pixel_values: torch.Tensor) -> torch.Tensor:
    """pixel_values fp32 [B, 3, S, S] -> image embedding as NHWC rows [B, (S/16)^2, out_chans], differentiable with respect to every
    parameter of the encoder: patch embedding, pos_embed, blocks (LayerNorm, qkv, windowed / global attention with the decomposed
    relative-position terms and their tables, proj, MLP), neck (1x1 conv, LayerNorm2d, 3x3 conv, LayerNorm2d)."""
    import torch.nn.functional as TF
    S, p = enc.img_size, enc.patch_size
    g = S // p
    N, D, C = g * g, enc.embed_dim, enc.out_chans
    B = pixel_values.shape[0]
    cols = ops.patch_im2col(_c(pixel_values.float()), S, p, F32, None, None)                   # [B*N, 3*p*p], a constant
    pe = enc.patch_embed.proj
    x = _apply_linear(cols, pe.weight.reshape(D, -1), pe.bias)
    if enc.pos_embed is not None:
        x = AddFn.apply(x, enc.pos_embed.reshape(N, D))
    for blk in enc.blocks:
        at, ws = blk.attn, blk.window_size
        heads, hd = at.num_heads, at.head_dim
        t = _ln(blk.norm1, x).reshape(B, g, g, D)
        if ws > 0:                                                                             # window_partition :243-264 (data movement; zero pad)
            pad = (ws - g % ws) % ws
            if pad:
                t = TF.pad(t, (0, 0, 0, pad, 0, pad))
            gp = g + pad
            t = t.reshape(B, gp // ws, ws, gp // ws, ws, D).permute(0, 1, 3, 2, 4, 5).contiguous().reshape(-1, ws, ws, D)
            Hh = ws
        else:
            Hh = g
        Bw, T = t.shape[0], Hh * Hh
        qkv = _apply_linear(t.reshape(Bw * T, D), at.qkv.weight, at.qkv.bias)
        bf_mode = BF16_LINEAR and at.qkv.weight.dtype == torch.bfloat16
        fused = None
        if FUSED_GLOBAL_FWD and bf_mode and RECOMPUTE_P and INPLACE_ATTN:
            # a block of a bf16 model: the forward VALUE from the inference path's kernels on the packed qkv (csrc/attention.hip vitglob_attn_kernel / win14r_attn_kernel / the tiled
            # kernel: bf16 operands, the decomposed rel-pos terms formed inside, fp32 softmax, bf16 probabilities and output) -- no score matrices in the forward.  Windowed blocks: the
            # kernel takes the UNpartitioned tokens (it gathers the windows and makes the pad tokens' k / v from the qkv bias itself); its output goes back into the padded-window
            # layout with zeros on the pad rows, whose outputs window_unpartition drops (and whose gradients are therefore zero) anyway
            with torch.no_grad():
                bf = torch.bfloat16
                qb = at.qkv.bias if at.qkv.bias is not None else torch.zeros(3 * D, device=x.device)
                qd = qkv.detach()
                if ws > 0:
                    qd = qd.reshape(B, gp // ws, gp // ws, ws, ws, 3 * D).permute(0, 1, 3, 2, 4, 5).reshape(B, gp, gp, 3 * D)[:, :g, :g]
                fused = ops.vit_attention(qd.to(bf).contiguous().reshape(B * N, 3 * D), at.rel_table("rh", at.rel_pos_h, Hh, bf), at.rel_table("rw", at.rel_pos_w, Hh, bf), at.cdt("qb", qb, bf),
                                          B, heads, hd, g, g, ws).float()
                if ws > 0:
                    fused = fused.reshape(B, g, g, D)
                    if gp > g:
                        fused = TF.pad(fused, (0, 0, 0, gp - g, 0, gp - g))
                    fused = fused.reshape(B, gp // ws, ws, gp // ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(Bw * T, D).contiguous()
        q, k, v = SplitFn.apply(qkv.reshape(Bw * T, 3, heads * hd), (1, 1, 1))
        def table(p):                                                                           # get_rel_pos's interpolation (:306-318) of a table of another length:
            if p.shape[0] == 2 * Hh - 1:                                                        # F.interpolate(mode="linear") = the resize kernel on [hd planes] x [1 x L] images;
                return p                                                                        # its adjoint (ResizeFn.backward) carries the gradient back to the stored rows
            return ResizeFn.apply(p.float().t().reshape(hd, 1, 1, p.shape[0]), (1, 2 * Hh - 1)).reshape(hd, 2 * Hh - 1).t()
        ar = torch.arange(Hh, device=x.device)
        idx = (ar[:, None] - ar[None, :] + (Hh - 1)).reshape(-1)                                # relative_coords :320-322
        Rh = GatherRowsFn.apply(table(at.rel_pos_h), idx).reshape(Hh, Hh, hd)
        Rw = GatherRowsFn.apply(table(at.rel_pos_w), idx).reshape(Hh, Hh, hd)
        q5 = q.reshape(Bw, Hh, Hh, heads, hd)                                                  # add_decomposed_rel_pos :325-361 (unscaled q)
        rel_h = BmmNTFn.apply(q5.permute(1, 0, 2, 3, 4).reshape(Hh, Bw * Hh * heads, hd), Rh)   # [qh, (b, qw, head), kh]
        rel_h = rel_h.reshape(Hh, Bw, Hh, heads, Hh).permute(1, 3, 0, 2, 4).reshape(Bw, heads, T, Hh)
        rel_w = BmmNTFn.apply(q5.permute(2, 0, 1, 3, 4).reshape(Hh, Bw * Hh * heads, hd), Rw)   # [qw, (b, qh, head), kw]
        rel_w = rel_w.reshape(Hh, Bw, Hh, heads, Hh).permute(1, 3, 2, 0, 4).reshape(Bw, heads, T, Hh)
        a = AttentionFn.apply(q, k, v, Bw, heads, heads, T, T, -1, None, rel_h, rel_w, Hh, bf_mode, fused)
        a = _apply_linear(a, at.proj.weight, at.proj.bias)
        if ws > 0:                                                                             # window_unpartition :267-290
            a = a.reshape(B, gp // ws, gp // ws, ws, ws, D).permute(0, 1, 3, 2, 4, 5).contiguous().reshape(B, gp, gp, D)
            if gp > g:
                a = a[:, :g, :g, :].contiguous()
        x = AddFn.apply(x, a.reshape(B * N, D))
        m = blk.mlp
        x = AddFn.apply(x, _linear(m.lin2, ActFn.apply(_linear(m.lin1, _ln(blk.norm2, x)), m.act_code)))
    n0, n1, n2, n3 = enc.neck[0], enc.neck[1], enc.neck[2], enc.neck[3]
    y = _ln(n1, _apply_linear(x, n0.weight.reshape(C, D), None))
    z = _apply_linear(Im2col3x3Fn.apply(y, B, g, g), n2.weight.permute(0, 2, 3, 1).reshape(C, 9 * C), None)
    return _ln(n3, z).reshape(B, N, C)


# ---------------------------------------------------------------------------------------------------------------------------------------
# The same graph behind the MODULES' own forward: in train() mode, with gradients enabled and something to differentiate, ImageEncoderViT /
# PromptEncoder / MaskDecoder / InternVLSAMModel.forward dispatch here, so that the reference's trainer (train_joint_v2.py:988-1100: model(...),
# model.vision_model(...), model.prompt_encoder(...), model.mask_decoder(...), then its own F.interpolate and calc_instance_loss) runs
# without edits.  NCHW <-> row conversions are torch data movement (permute / reshape); the arithmetic is the Functions above.
def wants_autograd(module: torch.nn.Module, *tensors) -> bool:
    """train() mode + autograd on + a parameter of the module or one of the given inputs to differentiate.  eval() models (what build_sam and
    app.py use) and calls under torch.no_grad() (the trainer's validation) stay on the inference kernels."""
    if not (module.training and torch.is_grad_enabled()):
        return False
    if any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in module.parameters())


def _rows_to_nchw(rows: torch.Tensor, B: int, h: int, w: int) -> torch.Tensor:
    return rows.reshape(B, h, w, -1).permute(0, 3, 1, 2)


def _nchw_to_rows(x: torch.Tensor) -> torch.Tensor:
    B, C, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * h * w, C)


def vision_forward(enc, pixel_values: torch.Tensor) -> torch.Tensor:
    """ImageEncoderViT.forward with gradients (image_encoder.py:106-117): [B, 3, S, S] -> [B, out_chans, S/16, S/16] fp32."""
    g = enc.img_size // enc.patch_size
    return _rows_to_nchw(vision_feature_rows(enc, pixel_values), pixel_values.shape[0], g, g)


def prompt_encoder_forward(pe, points, boxes, masks, llm_hidden_states):
    """PromptEncoder.forward with gradients (prompt_encoder.py:153-203): point prompts (+ the pad point), box prompts (the trainer forwards `boxes=`,
    train_joint_v2.py:975,1038,1057) and the LLM dense prompt or the no-mask embedding; -> (sparse [P, n, C], dense [P, C, h, w]) fp32."""
    if points is None and boxes is None and masks is None:
        raise NotImplementedError("the differentiable prompt encoder needs point, box or mask prompts")
    h, w = pe.image_embedding_size
    C = pe.embed_dim
    dev = pe.no_mask_embed.weight.device
    if masks is not None:                                                        # _embed_masks (prompt_encoder.py:105-108, 187-188): the dense prompt IS the downscaled mask
        dense = mask_downscaling_forward(pe, masks.to(dev))
        if points is None and boxes is None:
            return torch.empty((dense.shape[0], 0, C), dtype=F32, device=dev), dense
    coords = labels = bx = None
    if points is not None:
        coords, labels = points[0].to(dev).float().contiguous(), points[1].to(dev).to(torch.int32).contiguous()
    if boxes is not None:
        bx = boxes.to(dev).float().reshape(-1, 4).contiguous()                   # _embed_boxes, prompt_encoder.py:96-103
    P = coords.shape[0] if coords is not None else bx.shape[0]
    table = torch.cat([pe.not_a_point_embed.weight] + [e.weight for e in pe.point_embeddings], 0)
    sparse = SparseEmbedFn.apply(table, coords, labels, bx, pe.pe_layer.G(), pe.input_image_size)
    if masks is not None:
        pass                                                                     # (dense computed above: masks take precedence over the LLM dense prompt, as in the reference's if / else)
    elif llm_hidden_states is not None:
        x = llm_hidden_states
        n = x.shape[0]
        rows = _nchw_to_rows(_c(x) if not x.requires_grad else x.float())
        dense_rows = ScaleShiftFn.apply(LayerNormFn.apply(rows, None, None, 1e-5), pe.llm_scale_factor, pe.llm_bias)
        dense = _rows_to_nchw(dense_rows, n, h, w)
        if n != P:
            dense = dense.reshape(P, -1, h, w)   # same failure mode as the reference's reshape (:195-197)
    else:
        dense = _rows_to_nchw(BroadcastRowsFn.apply(pe.no_mask_embed.weight.reshape(1, C), P * h * w).reshape(P * h * w, C), P, h, w)
    return sparse, dense


def _conv_k2s2_rows(conv, x_nhwc: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d(kernel 2, stride 2) on an NHWC tensor: every 2x2 block of input pixels is one row (ky, kx, cin) of a Linear whose weight is the
    conv's viewed as [cout, (ky, kx, cin)] -> NHWC [P, H/2, W/2, cout].  The regrouping is torch data movement; autograd routes the gradients back."""
    P, H, W, Cin = x_nhwc.shape
    rows = x_nhwc.reshape(P, H // 2, 2, W // 2, 2, Cin).permute(0, 1, 3, 2, 4, 5).reshape(P * (H // 2) * (W // 2), 4 * Cin)
    wmat = conv.weight.permute(0, 2, 3, 1).reshape(conv.weight.shape[0], 4 * Cin)
    return _apply_linear(rows, wmat, conv.bias).reshape(P, H // 2, W // 2, -1)


def mask_downscaling_forward(pe, masks: torch.Tensor) -> torch.Tensor:
    """PromptEncoder._embed_masks with gradients (prompt_encoder.py:54-62, 105-108): Conv2d(1, c/4, 2, 2) -> LayerNorm2d -> GELU -> Conv2d(c/4, c, 2, 2) -> LayerNorm2d
    -> GELU -> Conv2d(c, embed_dim, 1) on masks [P, 1, 4h, 4w] -> [P, embed_dim, h, w] fp32; gradients reach every parameter of mask_downscaling (and the masks)."""
    md = pe.mask_downscaling
    x = masks.float().permute(0, 2, 3, 1)                                        # NHWC
    x = _conv_k2s2_rows(md[0], x)
    P, H, W, c1 = x.shape
    x = ActFn.apply(LayerNormFn.apply(x.reshape(-1, c1), md[1].weight, md[1].bias, md[1].eps), 1).reshape(P, H, W, c1)
    x = _conv_k2s2_rows(md[3], x)
    P, H, W, c2 = x.shape
    rows = ActFn.apply(LayerNormFn.apply(x.reshape(-1, c2), md[4].weight, md[4].bias, md[4].eps), 1)
    rows = _apply_linear(rows, md[6].weight.reshape(md[6].weight.shape[0], c2), md[6].bias)     # 1x1 convolution
    return _rows_to_nchw(rows, P, H, W)


def mask_decoder_forward(md, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output: bool):
    """MaskDecoder.forward with gradients (mask_decoder.py:71-149): -> (masks [P, 1 or 3, 4h, 4w], iou_pred [P, 1 or 3]) fp32."""
    B, C, h, w = image_embeddings.shape
    N = h * w
    sparse = sparse_prompt_embeddings if sparse_prompt_embeddings.dtype == F32 else sparse_prompt_embeddings.float()
    P = sparse.shape[0]
    if B != 1 and B != P:
        raise ValueError(f"image_embeddings batch {B} is incompatible with {P} prompts (reference repeat_interleave semantics)")
    out_tok = torch.cat([md.iou_token.weight, md.mask_tokens.weight], 0)
    tokens = torch.cat([BroadcastRowsFn.apply(out_tok, P), sparse], 1)
    img_rows = _nchw_to_rows(image_embeddings.float())                                     # [B*N, C]
    d = dense_prompt_embeddings
    if d.shape[0] != P:
        d = d.expand(P, C, h, w)
    src = AddFn.apply(_nchw_to_rows(d.float()), img_rows)                                  # repeat_interleave(image, P) + dense (:126-127)
    key_pe = _nchw_to_rows(image_pe[:1].detach().float()).contiguous()
    hs, keys = _two_way_transformer(md.transformer, src, key_pe, tokens)
    up0, ln, up1 = md.output_upscaling[0], md.output_upscaling[1], md.output_upscaling[3]
    c8 = C // 8
    u = _conv_transpose_k2s2(up0, keys)
    u = ActFn.apply(LayerNormFn.apply(u, ln.weight, ln.bias, ln.eps), 1)
    u = ActFn.apply(_conv_transpose_k2s2(up1, u), 1)
    nm = md.num_mask_tokens

    def mlp(m, x):
        for j, l in enumerate(m.layers):
            x = _linear(l, x)
            if j < m.num_layers - 1:
                x = ActFn.apply(x, 2)
        return x

    hyper = torch.stack([mlp(md.output_hypernetworks_mlps[i], hs[:, 1 + i, :]) for i in range(nm)], 1)
    m = HyperMasksFn.apply(hyper, u.reshape(P, N * 16, c8))
    m = m.reshape(P, nm, h, w, 2, 2, 2, 2).permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(P, nm, 4 * h, 4 * w)
    iou = mlp(md.iou_prediction_head, hs[:, 0, :])
    sl = slice(1, None) if multimask_output else slice(0, 1)                               # mask_decoder.py:100-105
    return m[:, sl, :, :], iou[:, sl]


def composite_forward(model, pixel_values, input_ids, attention_mask=None, labels=None, output_hidden_states=None):
    """InternVLSAMModel.forward with gradients (modeling_internvl_sam.py:106-224 as train_joint_v2.py:988-998 calls it): the vision model runs
    without gradients here, as in the reference (extract_feature, :243-244); mlp1 -> frozen LLM -> mlp2 is differentiable; `.loss` is the
    language-model loss of the reference's forward (differentiable: `lm_loss`); `.hidden_states` is the dense feature [B, 256, 64, 64] the segmentation
    branch continues from.  `image_flags` / `position_ids` of the reference's signature do not enter this path (the reference ignores image_flags too, :119-135)."""
    from .modeling.outputs import CausalLMOutputWithPast
    lm = model.language_model
    B, S = input_ids.shape
    with torch.no_grad():
        img_tok = model.vision_model.forward_tokens(pixel_values)                          # [B, 4096, 256] fp32
        g = int(math.sqrt(img_tok.shape[1]))
        image_embeddings = ops.transpose(img_tok, B, g * g, img_tok.shape[-1]).reshape(B, -1, g, g).to(model.dtype)
    hidden_img, hidden_all = llm_image_hidden(model, img_tok, input_ids, attention_mask, return_all=True)
    loss = None
    logits_fn = lambda: lm.lm_head(hidden_all.detach())
    if labels is not None:
        # `.loss` is differentiable, as in the reference: its trainer's other branch (train_joint_v2.py: masks is None or use_llm_hidden_states False) back-propagates
        # `outputs.loss` into mlp1 through the frozen LLM; in the segmentation branch the loss enters as 0 * loss (:1096) and its backward costs one head GEMM pair
        loss = lm_loss(lm, hidden_all, labels)
    hs = None
    if output_hidden_states:
        hs = _rows_to_nchw(dense_feature_rows(model, hidden_img), B, g, g)
    ret = CausalLMOutputWithPast(loss=loss, logits=None, logits_fn=logits_fn, past_key_values=None, hidden_states=hs, attentions=None)
    ret.image_embeddings = image_embeddings
    ret.image_tokens = img_tok
    ret.dense_feature_tokens = None
    return ret


def train_step_loss(model, pixel_values: torch.Tensor, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                    points: Tuple[torch.Tensor, torch.Tensor], gt_masks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The differentiable part of one step of train_joint_v2.py:943-1100 (batch 1, the reference's `loss = 0 * lm_loss + seg_loss`):
    model(...) with output_hidden_states (the vision features enter the LLM path under no_grad, modeling_internvl_sam.py:243-244), the second
    vision_model(pixel_values) call with gradients for the decoder, prompt encoder, mask decoder, upsample, BCE + Dice.  One vision-model
    forward serves both uses (same values; the LLM path takes it detached).  Returns (total, bce, dice); total.backward() fills the
    gradients of vision_model, mlp1, mlp2, prompt_encoder and mask_decoder parameters that require them."""
    rows = vision_feature_rows(model.vision_model, pixel_values)
    hidden = llm_image_hidden(model, rows.detach(), input_ids, attention_mask)
    return segmentation_loss(model, hidden, None, points, gt_masks, image_rows=rows)


class TrainStep(torch.nn.Module):
    """`train_step_loss` as a module, so that the trainer's `DistributedDataParallel(model, ...)` wrapping (train_joint_v2.py:1690-1700) keeps
    working: DDP hooks the parameters of `self.model` and all-reduces their gradients over RCCL while `backward()` runs; one process per GPU,
    one image per process and step, as the reference launches it (scripts/train_all_joint_v2.sh: torchrun --nproc_per_node).

        step = DistributedDataParallel(TrainStep(model), device_ids=[local_rank], find_unused_parameters=True)
        loss, bce, dice = step(pixel_values, input_ids, attention_mask, points, point_labels, masks);  loss.backward();  optimizer.step()

    (`find_unused_parameters=True`: the IoU head, the mask-input convolutions and the box-corner embeddings take no part in this loss, as in the
    reference's own step.)"""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, pixel_values, input_ids, attention_mask, points, point_labels, gt_masks):
        return train_step_loss(self.model, pixel_values, input_ids, attention_mask, (points, point_labels), gt_masks)
