"""Kernel-level parity: each HIP kernel (through the C ABI) vs the numpy oracle on seeded inputs.

fp32 mode must match to accumulation-order noise (the f32 MFMA is an exact fma chain); bf16 mode is compared with
a tolerance that reflects bf16 operand rounding (stated per test).
"""
import math

import numpy as np
import pytest
import torch

from oracle import ullsam_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"

GEMM_PERSIST_DEFAULT = 2   # ullsam_set_gemm_tuning(2, v): the library's default (csrc/gemm.hip g_persist); tests that switch it put it back


def T(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV).to(dtype).contiguous()


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.fixture(scope="module")
def ops():
    from ullsam_amd import ops as o
    return o


TOL = {torch.float32: 2e-4, torch.bfloat16: 6e-2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (128, 384, 768), (1, 128, 64), (515, 200, 256)])
def test_gemm_plain_and_epilogues(ops, dtype, M, N, K):
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    res = rng.standard_normal((M, N), dtype=np.float32)
    ad, wd = T(a, dtype), T(w, dtype)
    a_r, w_r = ad.float().cpu().numpy(), wd.float().cpu().numpy()  # operands as the kernel sees them
    ref = a_r @ w_r.T
    tol = 2e-4 if dtype == torch.float32 else 2e-2  # bf16: only output rounding / accumulation order remains
    y = ops.gemm(ad, wd, out_f32=True).cpu().numpy()
    assert err(y, ref) < tol
    y = ops.gemm(ad, wd, bias=T(bias), act=ops.ACT_GELU, residual=T(res), out_f32=True).cpu().numpy()
    assert err(y, O.gelu(ref + bias) + res) < tol
    y = ops.gemm(ad, wd, bias=T(bias), act=ops.ACT_RELU).float().cpu().numpy()
    assert err(y, np.maximum(ref + bias, 0)) < (tol if dtype == torch.float32 else 5e-2)
    # row-broadcast residual (pos_embed) + in-place residual update
    mod = max(1, M // 3)
    y = ops.gemm(ad, wd, residual=T(res[:mod]), res_row_mod=mod, out_f32=True).cpu().numpy()
    assert err(y, ref + res[np.arange(M) % mod]) < tol
    x = T(res)
    ops.gemm(ad, wd, bias=T(bias), residual=x, out_f32=True, out=x)
    assert err(x.cpu().numpy(), ref + bias + res) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_swiglu_pair(ops, dtype):
    rng = np.random.default_rng(5)
    M, I, K = 200, 256, 128
    x = rng.standard_normal((M, K), dtype=np.float32)
    w1 = (rng.standard_normal((I, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32)
    w3 = (rng.standard_normal((I, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32)
    from ullsam_amd.packing import pack_w13
    w13 = pack_w13(T(w1, dtype), T(w3, dtype))
    xd = T(x, dtype)
    y = ops.gemm(xd, w13, act=ops.ACT_SWIGLU, out_f32=True).cpu().numpy()
    xr = xd.float().cpu().numpy()
    ref = O.silu(xr @ T(w1, dtype).float().cpu().numpy().T) * (xr @ T(w3, dtype).float().cpu().numpy().T)
    assert y.shape == (M, I)
    assert err(y, ref) < (2e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("M,N,K,mode", [
    (1500, 512, 512, "bias_gelu"),        # 12 tiles, ragged M
    (1500, 512, 512, "res"),              # fp32 in-place residual, ragged M
    (16384, 4096, 512, "plain"),          # 1024 tiles = 4 rounds with a short K loop (16 stages)
    (16384, 1280, 5120, "res"),           # ViT lin2 shape on 256x256 tiles (320 tiles)
    (4324, 4096, 4096, "res"),            # LLM wo shape on 256x256 tiles (272 tiles, ragged M)
    (4324, 4096, 4096, "bias_relu"),
    (4324, 7168, 1024, "swiglu"),         # packed [gate | up] pairs, ragged M
    (16384, 1280, 768, "rowmod"),         # patch embedding: fp32 out + bias + row-broadcast residual (pos_embed)
    (2048, 256, 2304, "f32out"),          # neck 3x3: fp32 out, nothing else
    (1000, 768, 320, "plain"),            # K not a multiple of 128, ragged M
    (1001, 768, 256, "plain"),            # odd M: the LDS-staged epilogue
    (1000, 776, 256, "bias"),             # N not a multiple of 64: ragged last line group
    (512, 256, 256, "f32_gelu"),          # activation on an fp32 output: staged epilogue
])
def test_gemm_ring_kernel_256x256(ops, M, N, K, mode):
    """The ring kernel at its plain 256x256 tile (variant 6: four 32-deep LDS stages, two wave groups, epilogues straight from the
    accumulators where the layout allows, LDS-staged otherwise), forced by the variant switch, against torch's fp32 matmul of the same
    bf16-rounded operands; every epilogue, ragged M / N, run-to-run determinism."""
    from ullsam_amd import _lib
    from ullsam_amd.packing import pack_w13
    lib = _lib.load()
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.float() @ w.float().T
    F = torch.nn.functional
    try:
        lib.ullsam_set_gemm_variant(6)
        if mode == "plain":
            got, want, tol = ops.gemm(a, w).float(), ref, 3e-2
        elif mode == "bias":
            got, want, tol = ops.gemm(a, w, bias).float(), ref + bias, 3e-2
        elif mode == "bias_gelu":
            got, want, tol = ops.gemm(a, w, bias, act=ops.ACT_GELU).float(), F.gelu(ref + bias), 3e-2
        elif mode == "bias_relu":
            got, want, tol = ops.gemm(a, w, bias, act=ops.ACT_RELU).float(), torch.relu(ref + bias), 3e-2
        elif mode == "res":
            x = torch.randn(M, N, device=DEV, generator=g)
            want = ref + bias + x
            ops.gemm(a, w, bias, residual=x, out_f32=True, out=x)
            got, tol = x, 2e-3
        elif mode == "rowmod":
            r = torch.randn(4096, N, device=DEV, generator=g)
            got = ops.gemm(a, w, bias, residual=r, res_row_mod=4096, out_f32=True)
            want, tol = ref + bias + r.repeat(M // 4096, 1), 2e-3
        elif mode == "f32out":
            got, want, tol = ops.gemm(a, w, out_f32=True), ref, 2e-3
        elif mode == "f32_gelu":
            got, want, tol = ops.gemm(a, w, bias, act=ops.ACT_GELU, out_f32=True), F.gelu(ref + bias), 2e-3
        else:
            I = N // 2
            w13 = pack_w13(w[:I].contiguous(), w[I:].contiguous())
            got = ops.gemm(a, w13, act=ops.ACT_SWIGLU).float()
            want, tol = F.silu(ref[:, :I]) * ref[:, I:], 3e-2
        torch.cuda.synchronize()
        if mode == "plain":
            again = ops.gemm(a, w).float()
            lib.ullsam_set_gemm_variant(3)
            old = ops.gemm(a, w).float()
    finally:
        lib.ullsam_set_gemm_variant(0)
    d = (got - want).abs()
    assert got.shape == want.shape
    assert float(d.max()) < tol * max(1.0, float(want.abs().max()) / 4), (float(d.max()), float(want.abs().max()))
    if mode == "plain":   # run-to-run determinism and agreement with the two-buffer kernel on the same problem
        assert torch.equal(again, got)
        assert float((old - got).abs().max()) < 1e-2


@pytest.mark.parametrize("M,N,K,mode", [(4096, 1280, 1280, "res"), (2000, 960, 320, "bias_gelu"), (1234, 1000, 256, "plain"), (3000, 640, 1024, "res_mod")])
def test_gemm_256x320_kernel(ops, M, N, K, mode):
    """The 256x320-tile kernel (variant 8: a fifth column of sub-tiles per wave, so that widths of 1280 / 3840 give whole rounds of
    tiles): ragged M and N (also N not a multiple of 320), each epilogue it is dispatched for, against the fp32 matmul."""
    from ullsam_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.float() @ w.float().T
    try:
        lib.ullsam_set_gemm_variant(8)
        if mode == "plain":
            got, want, tol = ops.gemm(a, w).float(), ref, 3e-2
        elif mode == "bias_gelu":
            got, want, tol = ops.gemm(a, w, bias, act=ops.ACT_GELU).float(), torch.nn.functional.gelu(ref + bias), 3e-2
        elif mode == "res":
            x = torch.randn(M, N, device=DEV, generator=g)
            want = ref + bias + x
            ops.gemm(a, w, bias, residual=x, out_f32=True, out=x)
            got, tol = x, 2e-3
        else:
            pe = torch.randn(M // 2, N, device=DEV, generator=g)
            got, want, tol = ops.gemm(a, w, bias, residual=pe, res_row_mod=M // 2, out_f32=True), ref + bias + pe.repeat(2, 1), 2e-3
        torch.cuda.synchronize()
    finally:
        lib.ullsam_set_gemm_variant(0)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < tol * max(1.0, float(want.abs().max()) / 4)


@pytest.mark.parametrize("M,N,K,mode", [(4324, 4096, 1024, "res"), (4324, 2048, 512, "swiglu"), (1000, 700, 256, "bias_gelu"), (4352, 512, 320, "plain")])
def test_gemm_272x256_kernel(ops, M, N, K, mode):
    """The 272x256-tile kernel (variant 9: nine sub-tile rows in the upper wave row, eight in the lower -- 16 tile rows cover the
    4 x 1081 = 4324 prompt rows of the bench, so the LLM's GEMMs become whole rounds of tiles): ragged M / N, the SwiGLU and the fp32
    residual epilogues, against the fp32 matmul."""
    from ullsam_amd import _lib
    from ullsam_amd.packing import pack_w13
    lib = _lib.load()
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.float() @ w.float().T
    F = torch.nn.functional
    try:
        lib.ullsam_set_gemm_variant(9)
        if mode == "plain":
            got, want, tol = ops.gemm(a, w).float(), ref, 3e-2
        elif mode == "bias_gelu":
            got, want, tol = ops.gemm(a, w, bias, act=ops.ACT_GELU).float(), F.gelu(ref + bias), 3e-2
        elif mode == "res":
            x = torch.randn(M, N, device=DEV, generator=g)
            want = ref + bias + x
            ops.gemm(a, w, bias, residual=x, out_f32=True, out=x)
            got, tol = x, 2e-3
        else:
            I = N // 2
            got = ops.gemm(a, pack_w13(w[:I].contiguous(), w[I:].contiguous()), act=ops.ACT_SWIGLU).float()
            want, tol = F.silu(ref[:, :I]) * ref[:, I:], 3e-2
        torch.cuda.synchronize()
    finally:
        lib.ullsam_set_gemm_variant(0)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < tol * max(1.0, float(want.abs().max()) / 4)


@pytest.mark.parametrize("M,N,K,mode", [(1081, 4096, 14336, "res"), (1081, 4096, 5120, "bias"), (1081, 4096, 6144, "plain"), (1081, 4096, 28672, "f32"),
                                        (4096, 1280, 5120, "res"), (2049, 2048, 8192, "bias"), (1081, 4096, 1024, "plain")])
def test_gemm_split_k_ring_against_float64_and_the_128_tile_kernel(ops, M, N, K, mode):
    """Launches of few ring tiles under a long sum (csrc/gemm.hip launch_gemm_ring_splitk: 1081 x 4096 outputs = a batch-1 prefill's wo / w2 and the frozen LLM's products of a
    training step; the AMG encoder's lin2 at one image) run as up to 8 K ranges of the ring kernel side by side, fp32 planes in the workspace added in order by splitk_finish_kernel
    (+ bias / residual, bf16 or fp32 out).  Against float64 on the bf16-rounded operands, against the 128x128 kernel that takes these shapes without the caller's ULLSAM_ACT_SPLITK_OK (the default; ullsam_set_gemm_tuning(3, 2) forces the split form),
    and twice (the sums are ordered: bit-equal).  Odd M, the 256x320 shape, in-place residual; a K without ranges of >= 1280 (the last case) stays on the old path."""
    from ullsam_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    x0 = torch.randn(M, N, device=DEV, generator=g)
    ref = (a.double() @ w.double().T)

    def run():
        if mode == "plain":
            return ops.gemm(a, w).double(), ref, 2e-2
        if mode == "bias":
            return ops.gemm(a, w, bias).double(), ref + bias.double(), 2e-2
        if mode == "f32":
            return ops.gemm(a, w, bias, out_f32=True).double(), ref + bias.double(), 1e-4
        x = x0.clone()
        ops.gemm(a, w, bias, residual=x, out_f32=True, out=x)
        return x.double(), ref + bias.double() + x0.double(), 1e-4
    try:
        lib.ullsam_set_gemm_tuning(3, 1)          # the default: a plain ops.gemm call does not carry ULLSAM_ACT_SPLITK_OK -> the one-launch kernel
        old, want, tol = run()
        lib.ullsam_set_gemm_tuning(3, 2)          # wherever the plan fits
        got, _, _ = run()
        again, _, _ = run()
        torch.cuda.synchronize()
    finally:
        lib.ullsam_set_gemm_tuning(3, 1)
    if mode == "plain" and K != 1024:             # the caller's flag does the same as the forced mode
        assert torch.equal(ops.gemm(a, w, splitk_ok=True).double(), got) and not torch.equal(got, old)
    scale = max(1.0, float(want.abs().max()) / 4)
    assert float((got - want).abs().max()) < tol * scale, float((got - want).abs().max())
    assert float((got - old).abs().max()) < 2 * tol * scale
    assert torch.equal(again, got)
    if K == 1024:
        assert torch.equal(got, old)     # no cut into ranges of >= 1280 exists: ring_splitk_plan declines, the launch is the old one
    mean_err_new, mean_err_old = float((got - want).abs().mean()), float((old - want).abs().mean())
    assert mean_err_new <= 1.05 * mean_err_old + 1e-7, (mean_err_new, mean_err_old)


@pytest.mark.parametrize("variant,M,N,K,mode", [(9, 4324, 8192, 384, "swiglu"), (9, 8704, 4096, 512, "res"), (8, 8192, 3840, 384, "bias_gelu"), (8, 8000, 3840, 640, "res_mod"),
                                                (6, 8192, 4096, 384, "bias"), (6, 6000, 6144, 1024, "plain")])
@pytest.mark.parametrize("persist", [1, 2, 4])
def test_gemm_persistent_ring_equals_the_one_tile_kernel(ops, variant, M, N, K, mode, persist):
    """The persistent forms of the ring kernel (csrc/gemm_ring8p.h: ullsam_set_gemm_tuning(2, 1) = workgroups walk tiles with the LDS ring kept full across tile borders,
    4 = the same with ONE barrier per stage) on launches of more than one round of tiles, every direct epilogue, ragged M: same MFMA order and same epilogue arithmetic as the
    one-tile-per-workgroup kernel, so the outputs must be EQUAL bit for bit (and both are checked against the fp32 matmul)."""
    from ullsam_amd import _lib
    from ullsam_amd.packing import pack_w13
    lib = _lib.load()
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K + variant)
    a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.float() @ w.float().T
    F = torch.nn.functional
    x0 = torch.randn(M, N, device=DEV, generator=g)
    pe = torch.randn(M // 2, N, device=DEV, generator=g)

    def run():
        if mode == "plain":
            return ops.gemm(a, w).float(), ref, 3e-2
        if mode == "bias":
            return ops.gemm(a, w, bias).float(), ref + bias, 3e-2
        if mode == "bias_gelu":
            return ops.gemm(a, w, bias, act=ops.ACT_GELU).float(), F.gelu(ref + bias), 3e-2
        if mode == "res":
            x = x0.clone()
            ops.gemm(a, w, bias, residual=x, out_f32=True, out=x)
            return x, ref + bias + x0, 2e-3
        if mode == "res_mod":
            return ops.gemm(a, w, bias, residual=pe, res_row_mod=M // 2, out_f32=True), ref + bias + pe.repeat(2, 1), 2e-3
        I = N // 2
        return ops.gemm(a, pack_w13(w[:I].contiguous(), w[I:].contiguous()), act=ops.ACT_SWIGLU).float(), F.silu(ref[:, :I]) * ref[:, I:], 3e-2
    try:
        lib.ullsam_set_gemm_variant(variant)
        lib.ullsam_set_gemm_tuning(2, 0)
        base, want, tol = run()
        lib.ullsam_set_gemm_tuning(2, persist)
        got, _, _ = run()
        again, _, _ = run()
        torch.cuda.synchronize()
    finally:
        lib.ullsam_set_gemm_variant(0)
        lib.ullsam_set_gemm_tuning(2, GEMM_PERSIST_DEFAULT)
    assert float((base - want).abs().max()) < tol * max(1.0, float(want.abs().max()) / 4)
    assert torch.equal(got, base), int((got != base).sum())
    assert torch.equal(again, got)


@pytest.mark.parametrize("persist", [1, 2])
@pytest.mark.parametrize("B,S,KVH,G,K,variant", [(4, 1081, 8, 4, 4096, 0), (2, 1500, 8, 4, 512, 9), (3, 1200, 8, 4, 1024, 6)])
def test_rope_gemm_persistent_ring_equals_the_one_tile_kernel(ops, B, S, KVH, G, K, variant, persist):
    """The wqkv GEMM with the head split + RoPE + KV-cache append in its epilogue (modeling_internlm2.py:359-388) on the persistent ring kernel (EMODE 1: 208- / 272- / 256-row
    tiles; the first case is the bench's launch: 504 tiles of 208x256) against the one-tile kernel: q, the appended K / V rows and the untouched cache rows EQUAL bit for bit."""
    from ullsam_amd import _lib
    lib = _lib.load()
    hd = 128
    g = torch.Generator(device=DEV); g.manual_seed(B * S + K)
    x = torch.randn(B * S, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(KVH * (G + 2) * hd, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    pos = ((torch.arange(S, device=DEV, dtype=torch.int32)[None] + 3 * torch.arange(B, device=DEV, dtype=torch.int32)[:, None]) % (S + 5)).contiguous()
    cos, sin = O.rope_tables(hd, S + 8, 1e6)
    cos, sin = T(cos), T(sin)
    cap, p0 = S + 6, 2
    res = {}
    try:
        lib.ullsam_set_gemm_variant(variant)
        for pz in (0, persist):
            lib.ullsam_set_gemm_tuning(2, pz)
            kc = torch.full((B, KVH, cap, hd), 0.25, dtype=torch.bfloat16, device=DEV); vc = torch.full_like(kc, -0.5)
            q = ops.gemm_qkv_rope(x, w, None, kc, vc, pos, cos, sin, B, S, KVH, G, p0)
            res[pz] = (q, kc, vc)
        torch.cuda.synchronize()
    finally:
        lib.ullsam_set_gemm_variant(0)
        lib.ullsam_set_gemm_tuning(2, GEMM_PERSIST_DEFAULT)
    for a, b in zip(res[0], res[persist]):
        assert torch.equal(a, b), int((a != b).sum())
    assert float((res[0][1][:, :, :p0] - 0.25).abs().max()) == 0 and float(res[0][0].float().abs().max()) > 0


def _e4m3_decode(u8: np.ndarray) -> np.ndarray:
    """OCP e4m3fn bytes -> float32 (the tests' own decoder: sign, 4-bit exponent bias 7, 3-bit mantissa, subnormals, 0x7f = NaN)."""
    u = u8.astype(np.int32)
    s_ = np.where(u & 0x80, -1.0, 1.0)
    e = (u >> 3) & 0xF
    m = u & 7
    v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e - 7.0))
    return (s_ * v).astype(np.float32)


@pytest.mark.parametrize("M,N,K,act", [(300, 256, 256, 0), (2048, 768, 1280, 1), (1000, 512, 128, 2)])
def test_fp8_rows_and_gemm(ops, M, N, K, act):
    """fp8 (OCP e4m3) path of BASELINE configs[4]: the row quantiser (with and without the fused LayerNorm) against a numpy
    re-statement, and the block-scaled-MFMA GEMM against an fp32 matmul of the DEQUANTISED operands (so only accumulation order and
    the bf16 output rounding remain)."""
    rng = np.random.default_rng(M + K)
    x = (rng.standard_normal((M, K), dtype=np.float32) * 2 + 0.3).astype(np.float32)
    lw = rng.standard_normal(K, dtype=np.float32); lb = rng.standard_normal(K, dtype=np.float32) * 0.1
    w = (rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    # quantiser: scale = amax / 448, values within half an e4m3 ulp of y / scale
    q, sc = ops.rows_fp8(T(x), T(lw), T(lb), 1e-6)
    y = O.layer_norm(x, lw, lb, 1e-6)
    sc_ref = np.abs(y).max(-1) / 448.0
    assert err(sc.cpu().numpy(), sc_ref) < 1e-6 * float(sc_ref.max()) + 1e-9
    deq = _e4m3_decode(q.cpu().numpy()) * sc.cpu().numpy()[:, None]
    ulp = np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(y / sc_ref[:, None]), 2.0 ** -6))) - 3), 2.0 ** -9)  # e4m3 spacing at |v|
    assert (np.abs(deq - y) <= (0.5 * ulp + 1e-6) * sc_ref[:, None] * 1.001).all()
    qw, sw = ops.rows_fp8(T(w))                                         # weights: no norm
    wdeq = _e4m3_decode(qw.cpu().numpy()) * sw.cpu().numpy()[:, None]
    assert err(sw.cpu().numpy(), np.abs(w).max(-1) / 448.0) < 1e-9 and np.abs(wdeq - w).max() <= np.abs(w).max() / 16
    # GEMM
    ref = deq @ wdeq.T + bias
    if act == 1:
        ref = O.gelu(ref)
    elif act == 2:
        ref = np.maximum(ref, 0)
    got = ops.gemm_fp8(q, sc, qw, sw, T(bias), act=act).float().cpu().numpy()
    assert got.shape == (M, N)
    assert err(got, ref) < 2e-2 * max(1.0, float(np.abs(ref).max()) / 4)
    got32 = ops.gemm_fp8(q, sc, qw, sw, T(bias), act=act, out_dtype=torch.float32).cpu().numpy()
    assert err(got32, ref) < 2e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("D", [64, 256, 768, 1280, 4096])
def test_norms(ops, D):
    rng = np.random.default_rng(D)
    x = rng.standard_normal((37, D), dtype=np.float32) * 2 + 0.3
    w = rng.standard_normal(D, dtype=np.float32)
    b = rng.standard_normal(D, dtype=np.float32)
    y = ops.norm(T(x), T(w), T(b), 1e-6, torch.float32).cpu().numpy()
    assert err(y, O.layer_norm(x, w, b, 1e-6)) < 2e-5
    y = ops.norm(T(x), T(w), None, 1e-5, torch.float32, rms=True).cpu().numpy()
    assert err(y, O.rms_norm(x, w, 1e-5)) < 2e-5
    y = ops.norm(T(x), None, None, 1e-5, torch.float32, post_scale=T(np.array([0.1], np.float32)),
                 post_shift=T(np.array([0.05], np.float32))).cpu().numpy()
    assert err(y, O.layer_norm(x, None, None, 1e-5) * 0.1 + 0.05) < 2e-5
    yb = ops.norm(T(x, torch.bfloat16), T(w), T(b), 1e-6, torch.bfloat16).float().cpu().numpy()
    xb = T(x, torch.bfloat16).float().cpu().numpy()
    assert err(yb, O.layer_norm(xb, w, b, 1e-6)) < 4e-2


@pytest.mark.parametrize("M,N,K", [(448, 256, 256), (64, 32, 256), (17, 4, 256), (100, 2048, 256), (448, 256, 2048), (30, 130, 128), (449, 128, 128), (70, 64, 1024), (33, 32, 512)])
def test_skinny_linear(ops, M, N, K):
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) / np.float32(math.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32)
    r = rng.standard_normal((M, N), dtype=np.float32)
    wt = T(np.ascontiguousarray(w.T))
    assert err(ops.skinny_linear(T(x), wt, T(b)).cpu().numpy(), x @ w.T + b) < 1e-5
    assert err(ops.skinny_linear(T(x), wt, T(b), ops.ACT_RELU, T(r)).cpu().numpy(), np.maximum(x @ w.T + b, 0) + r) < 1e-5
    assert err(ops.skinny_linear(T(x), wt, None, ops.ACT_GELU).cpu().numpy(), O.gelu(x @ w.T)) < 1e-5
    assert err(ops.small_linear(T(x), T(w), T(b)).cpu().numpy(), x @ w.T + b) < 1e-5
    # the FMA kernel behind the switch (N % 32 == 0 and K in {128 .. 2048} take the exact-fp32 MFMA kernel by default): same results to fp32 rounding of the k sum
    from ullsam_amd import _lib
    lib = _lib.load()
    new = ops.skinny_linear(T(x), wt, T(b), ops.ACT_RELU, T(r))
    old_sw = lib.ullsam_set_skinny_linear_mfma(0)
    try:
        fma = ops.skinny_linear(T(x), wt, T(b), ops.ACT_RELU, T(r))
    finally:
        lib.ullsam_set_skinny_linear_mfma(old_sw)
    assert err(fma.cpu().numpy(), np.maximum(x @ w.T + b, 0) + r) < 1e-5
    assert float((new - fma).abs().max()) < 1e-5 * max(1.0, float(fma.abs().max()))


@pytest.mark.parametrize("D", [2048, 4096, 3072])
def test_norm_long_rows_workgroup_per_row(ops, D):
    """Rows of >= 2048 elements, >= 1024 of them (the LLM's RMSNorm): one workgroup per row, statistics across waves through LDS."""
    rng = np.random.default_rng(D)
    x = rng.standard_normal((1500, D), dtype=np.float32) * 2 + 0.3
    w = rng.standard_normal(D, dtype=np.float32)
    b = rng.standard_normal(D, dtype=np.float32)
    assert err(ops.norm(T(x), T(w), None, 1e-5, torch.float32, rms=True).cpu().numpy(), O.rms_norm(x, w, 1e-5)) < 2e-5
    assert err(ops.norm(T(x), T(w), T(b), 1e-6, torch.float32).cpu().numpy(), O.layer_norm(x, w, b, 1e-6)) < 2e-5
    got = ops.norm(T(x), T(w), None, 1e-5, torch.bfloat16, rms=True).float().cpu().numpy()
    assert err(got, O.rms_norm(x, w, 1e-5)) < 4e-2
    from ullsam_amd import _lib
    lib = _lib.load()
    try:  # the wave-per-row kernel gives the same numbers up to summation order
        lib.ullsam_set_norm_variant(1)
        alt = ops.norm(T(x), T(w), None, 1e-5, torch.float32, rms=True).cpu().numpy()
    finally:
        lib.ullsam_set_norm_variant(0)
    assert err(alt, O.rms_norm(x, w, 1e-5)) < 2e-5


@pytest.mark.parametrize("D", [32, 64, 48])
def test_norm_narrow_rows(ops, D):
    """Many short rows (LayerNorm2d + GELU of the decoder's upscaling path): 16 lanes per row."""
    rng = np.random.default_rng(D)
    x = rng.standard_normal((5003, D), dtype=np.float32) * 2 + 0.3
    w = rng.standard_normal(D, dtype=np.float32)
    b = rng.standard_normal(D, dtype=np.float32)
    ref = O.layer_norm(x, w, b, 1e-6)
    assert err(ops.norm(T(x), T(w), T(b), 1e-6, torch.float32).cpu().numpy(), ref) < 2e-5
    got = ops.norm(T(x), T(w), T(b), 1e-6, torch.bfloat16, act=ops.ACT_GELU).float().cpu().numpy()
    assert err(got, O.gelu(ref)) < 4e-2
    assert err(ops.norm(T(x), T(w), None, 1e-5, torch.float32, rms=True).cpu().numpy(), O.rms_norm(x, w, 1e-5)) < 2e-5


def _vit_attn_case(ops, dtype, heads, hd, grid, window, B=2, seed=0):
    D = heads * hd
    rng = np.random.default_rng(seed)
    xn = rng.standard_normal((B, grid, grid, D), dtype=np.float32)
    n = window if window > 0 else grid
    P = {"qkv.weight": (rng.standard_normal((3 * D, D), dtype=np.float32) / math.sqrt(D)).astype(np.float32),
         "qkv.bias": rng.standard_normal(3 * D, dtype=np.float32) * 0.3,
         "rel_pos_h": rng.standard_normal((2 * n - 1, hd), dtype=np.float32) * 0.1,
         "rel_pos_w": rng.standard_normal((2 * n - 1, hd), dtype=np.float32) * 0.1,
         "proj.weight": np.eye(D, dtype=np.float32), "proj.bias": np.zeros(D, np.float32)}
    # oracle: exactly Block.forward's attention branch (window_partition -> Attention -> window_unpartition)
    if window > 0:
        xw, pad_hw = O.window_partition(xn, window)
        ref = O.window_unpartition(O.vit_attention(xw, P, "", heads), window, pad_hw, (grid, grid))
    else:
        ref = O.vit_attention(xn, P, "", heads)
    qkv = O.linear(xn.reshape(-1, D), P["qkv.weight"], P["qkv.bias"])  # UNPADDED tokens
    out = ops.vit_attention(T(qkv, dtype), T(P["rel_pos_h"], dtype), T(P["rel_pos_w"], dtype), T(P["qkv.bias"], dtype),
                            B, heads, hd, grid, grid, window)
    return err(out.float().cpu().numpy().reshape(ref.shape), ref), float(np.abs(ref).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("heads,hd,grid,window", [(2, 64, 10, 7), (2, 64, 10, 0), (2, 80, 16, 14), (1, 80, 12, 0),
                                                  (2, 64, 64, 14), (1, 64, 64, 0),
                                                  # the instantiations the bench runs (ViT-H: 16 heads x 80 on SAM's 64x64 grid):
                                                  # flash_attn_kernel<bf16,80,VIT_GLOBAL,4,FAST64> and win14_attn_kernel<80> on 25 windows/image
                                                  (16, 80, 64, 0), (16, 80, 64, 14)])
def test_vit_attention(ops, dtype, heads, hd, grid, window):
    e, scale = _vit_attn_case(ops, dtype, heads, hd, grid, window)
    assert e < (1e-4 if dtype == torch.float32 else 3e-2) * max(1.0, scale), (e, scale)


@pytest.mark.parametrize("B,heads,qscale", [(4, 16, 1.0), (1, 3, 1.0), (2, 4, 6.0)])
def test_vit_global_attention_dma_kernel_against_the_tiled_kernel_and_float64(ops, B, heads, qscale):
    """vitglob_attn_kernel (bf16, head_dim 80, 64 x 64 grid: K / V by LDS-DMA in 32-key stages into swizzled rings, the two waves of a SIMD half an
    interval apart, the rel_w term in the MFMA's initial accumulator, the O rescale skipped when no maximum moved) against the tiled
    flash_attn_kernel<bf16, 80, VIT_GLOBAL, 8, FAST64> (attention variant 12) on the same operands: same MFMA order; the exponent's argument is
    rounded once instead of twice and the running maximum follows the true one lazily (probabilities up to 2^5 instead of 1 between rescales), so
    the outputs agree to bf16 rounding noise -- the bench's shape, a pair count that is not a multiple of the 8 XCDs, and queries scaled up so that
    the maximum keeps moving by more than the lazy threshold (the rescale branch is taken again and again) -- and both agree with a float64 softmax
    including the decomposed rel-pos bias (image_encoder.py:325-361), the new kernel no worse than the tiled one."""
    from ullsam_amd import _lib
    lib = _lib.load()
    hd, grid, D = 80, 64, heads * 80
    g = torch.Generator(device=DEV); g.manual_seed(B * 100 + heads)
    qkv = torch.randn(B * grid * grid, 3 * D, device=DEV, generator=g)
    qkv[:, :D] *= qscale
    qkv = qkv.bfloat16()
    rh = (torch.randn(2 * grid - 1, hd, device=DEV, generator=g) * 0.1).bfloat16()
    rw = (torch.randn(2 * grid - 1, hd, device=DEV, generator=g) * 0.1).bfloat16()
    bias = torch.zeros(3 * D, device=DEV).bfloat16()
    new = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, 0)
    try:
        lib.ullsam_set_attn_variant(12)
        old = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, 0)
    finally:
        lib.ullsam_set_attn_variant(0)
    torch.cuda.synchronize()
    dn = (new.float() - old.float()).abs()
    assert float(dn.max()) <= 2.0 ** -6 * max(1.0, float(old.float().abs().max())), float(dn.max())     # within two bf16 rounding steps of the largest output
    # float64 reference of one (image, head) pair on the operands as the kernel sees them
    b, h = B - 1, heads - 1
    t = qkv.double().reshape(B, grid * grid, 3, heads, hd)
    q, k, v = t[b, :, 0, h], t[b, :, 1, h], t[b, :, 2, h]
    idx = torch.arange(grid, device=DEV)
    rel = idx[:, None] - idx[None, :] + grid - 1                          # get_rel_pos: q - k + (G - 1)
    Rh, Rw = rh.double()[rel], rw.double()[rel]                           # [q coord, k coord, hd]
    qg = q.reshape(grid, grid, hd)
    bh = torch.einsum("hwc,hkc->hwk", qg, Rh)                             # [qh, qw, kh]
    bw = torch.einsum("hwc,wkc->hwk", qg, Rw)                             # [qh, qw, kw]
    sc = (q @ k.T) / math.sqrt(hd) + (bh[:, :, :, None] + bw[:, :, None, :]).reshape(grid * grid, grid * grid)
    ref = torch.softmax(sc, -1) @ v
    got = new.double().reshape(B, grid * grid, heads, hd)[b, :, h]
    got_old = old.double().reshape(B, grid * grid, heads, hd)[b, :, h]
    e_new, e_old = float((got - ref).abs().max()), float((got_old - ref).abs().max())
    print(f"vit global attention vs float64: LDS-DMA kernel {e_new:.3e}, tiled kernel {e_old:.3e} (mean {float((got - ref).abs().mean()):.2e} / {float((got_old - ref).abs().mean()):.2e})")
    assert e_new < 3e-2 * max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().mean()) <= 1.15 * float((got_old - ref).abs().mean()) + 1e-6


@pytest.mark.parametrize("B,heads,grid,qscale", [(4, 16, 64, 1.0), (1, 3, 16, 1.0), (2, 2, 30, 1.0), (1, 5, 64, 5.0)])
def test_vit_window_attention_row_group_kernel_against_the_block_kernel_and_float64(ops, B, heads, grid, qscale):
    """win14r_attn_kernel (round 5: a query group and a key tile are one 14-token window row on 16x16x32 MFMAs, all 14 score tiles before ONE softmax,
    rel_w in the MFMA's initial accumulator, rel_h in the exponent's offset, K / V gathered by LDS-DMA into plain 160-byte rows) against round 1's
    win14_attn_kernel (attention variant 13: 32-query groups, seven 32-key blocks, online softmax) on the same operands -- the bench's shape, a grid of
    2 x 2 windows of which three are padded (16 -> 28: pad tokens are live keys = qkv.bias, image_encoder.py:243-264), a 30 x 30 grid (3 x 3 windows, 12 pad
    rows / columns, window rows wholly outside the image), a head count that is not a multiple of the 8 XCDs, queries scaled up -- and against a float64
    softmax with the decomposed rel-pos bias from the UNSCALED q (image_encoder.py:325-361) on one (image, window, head) incl. a padded window."""
    from ullsam_amd import _lib
    lib = _lib.load()
    hd, W, D = 80, 14, heads * 80
    g = torch.Generator(device=DEV); g.manual_seed(B * 1000 + heads * 10 + grid)
    qkv = torch.randn(B * grid * grid, 3 * D, device=DEV, generator=g)
    qkv[:, :D] *= qscale
    qkv = qkv.bfloat16()
    rh = (torch.randn(2 * W - 1, hd, device=DEV, generator=g) * 0.1).bfloat16()
    rw = (torch.randn(2 * W - 1, hd, device=DEV, generator=g) * 0.1).bfloat16()
    bias = (torch.randn(3 * D, device=DEV, generator=g) * 0.3).bfloat16()
    new = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, W)
    try:
        lib.ullsam_set_attn_variant(13)
        old = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, W)
    finally:
        lib.ullsam_set_attn_variant(0)
    torch.cuda.synchronize()
    assert torch.isfinite(new.float()).all()
    dn = (new.float() - old.float()).abs()
    assert float(dn.max()) <= 2.0 ** -6 * max(1.0, float(old.float().abs().max())), float(dn.max())     # within two bf16 rounding steps of the largest output
    # float64 reference of the LAST window (bottom right: padded whenever the grid is not a multiple of 14) of the last image / head
    nw = (grid + W - 1) // W
    b, h, wy, wx = B - 1, heads - 1, nw - 1, nw - 1
    t = qkv.double().reshape(B, grid, grid, 3, heads, hd)
    win = bias.double().reshape(3, heads, hd)[:, h][None, None].repeat(W, W, 1, 1)          # pad tokens: q = k = v = qkv.bias
    ys, xs = min(W, grid - wy * W), min(W, grid - wx * W)
    win[:ys, :xs] = t[b, wy * W:wy * W + ys, wx * W:wx * W + xs, :, h]
    q, k, v = (win[:, :, i].reshape(W * W, hd) for i in range(3))
    idx = torch.arange(W, device=DEV)
    rel = idx[:, None] - idx[None, :] + W - 1
    Rh, Rw = rh.double()[rel], rw.double()[rel]
    qg = q.reshape(W, W, hd)
    bh = torch.einsum("hwc,hkc->hwk", qg, Rh)
    bw = torch.einsum("hwc,wkc->hwk", qg, Rw)
    sc = (q @ k.T) / math.sqrt(hd) + (bh[:, :, :, None] + bw[:, :, None, :]).reshape(W * W, W * W)
    ref = (torch.softmax(sc, -1) @ v).reshape(W, W, hd)[:ys, :xs]
    pick = lambda o: o.double().reshape(B, grid, grid, heads, hd)[b, wy * W:wy * W + ys, wx * W:wx * W + xs, h]
    e_new, e_old = (pick(new) - ref).abs(), (pick(old) - ref).abs()
    print(f"vit window attention vs float64: row-group kernel max {float(e_new.max()):.3e} mean {float(e_new.mean()):.2e}; block kernel max {float(e_old.max()):.3e} mean {float(e_old.mean()):.2e}")
    assert float(e_new.max()) < 3e-2 * max(1.0, float(ref.abs().max()))
    assert float(e_new.mean()) <= 1.15 * float(e_old.mean()) + 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,H,KVH,pad", [(2, 70, 2, 1, 9), (1, 200, 4, 2, 0), (1, 1081, 2, 2, 0)])
def test_rope_and_causal_attention(ops, dtype, B, S, H, KVH, pad):
    hd, G = 128, H // KVH
    rng = np.random.default_rng(S)
    qkv = rng.standard_normal((B, S, KVH, G + 2, hd), dtype=np.float32)
    mask = np.ones((B, S), np.int64)
    if pad:
        mask[-1, :pad] = 0
    pos = np.arange(S, dtype=np.int64)[None].repeat(B, 0)
    cos, sin = O.rope_tables(hd, S, 1e6)
    qd = T(qkv.reshape(B * S, -1), dtype)
    qr = qd.float().cpu().numpy().reshape(qkv.shape)
    q = qr[..., :G, :].reshape(B, S, H, hd).transpose(0, 2, 1, 3)
    k = qr[..., -2, :].transpose(0, 2, 1, 3)
    v = qr[..., -1, :].transpose(0, 2, 1, 3)
    c, s = cos[pos][:, None], sin[pos][:, None]
    q = q * c + O._rotate_half(q) * s
    k = k * c + O._rotate_half(k) * s
    a = np.matmul(q, np.repeat(k, G, 1).transpose(0, 1, 3, 2)) / np.float32(math.sqrt(hd)) + O.decoder_mask(mask, S, 0)
    ref = np.matmul(O.softmax(a.astype(np.float32)), np.repeat(v, G, 1)).transpose(0, 2, 1, 3).reshape(B, S, H * hd)
    kc = torch.zeros((B, KVH, S + 3, hd), dtype=dtype, device=DEV)
    vc = torch.zeros_like(kc)
    qo = ops.rope_split(qd, kc, vc, T(pos.astype(np.int32), torch.int32), T(cos), T(sin), B, S, KVH, G, hd, 0)
    assert err(kc[:, :, :S].float().cpu().numpy(), k) < (1e-5 if dtype == torch.float32 else 3e-2)
    out = ops.causal_attention(qo, kc, vc, T(mask.astype(np.int32), torch.int32) if pad else None, B, H, KVH, hd, S, S, 0)
    got = out.float().cpu().numpy().reshape(B, S, H * hd)
    valid = mask.astype(bool)
    assert err(got[valid], ref[valid]) < (2e-4 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize("B,Sq,past,H,KVH,pad", [(4, 1081, 0, 32, 8, 13), (2, 70, 0, 2, 1, 9), (1, 200, 0, 4, 2, 0), (1, 129, 0, 2, 2, 0),
                                                  (3, 64, 0, 4, 4, 5), (2, 300, 77, 8, 2, 20), (1, 1, 500, 4, 4, 0)])
def test_causal_attention_dma_kernel_is_bit_identical_to_the_tiled_kernel(ops, B, Sq, past, H, KVH, pad):
    """causal128_attn_kernel (bf16, head_dim 128: K/V tiles by LDS-DMA into a double-buffered, XOR-swizzled LDS image) against the tiled
    flash_attn_kernel (attention variant 11) on the same operands: same MFMA order, same mask arithmetic, so the outputs must be equal bit for
    bit -- ragged query blocks, left padding, cached keys before the first query (q_pos0 > 0), a single query, the bench's shape -- and both
    agree with a float64 softmax of the reference's additive finfo.min masks (modeling_internlm2.py:96-125)."""
    from ullsam_amd import _lib
    lib = _lib.load()
    hd, G, Sk = 128, H // KVH, past + Sq
    g = torch.Generator(device=DEV); g.manual_seed(Sq + past + H)
    q = torch.randn(B * Sq, H * hd, device=DEV, generator=g).bfloat16()
    cap = Sk + 5
    kc = torch.randn(B, KVH, cap, hd, device=DEV, generator=g).bfloat16()
    vc = torch.randn(B, KVH, cap, hd, device=DEV, generator=g).bfloat16()
    mask = torch.ones(B, Sk, dtype=torch.int32, device=DEV)
    if pad:
        mask[-1, :pad] = 0
    km = mask if pad else None
    new = ops.causal_attention(q, kc, vc, km, B, H, KVH, hd, Sq, Sk, past)
    try:
        lib.ullsam_set_attn_variant(11)
        old = ops.causal_attention(q, kc, vc, km, B, H, KVH, hd, Sq, Sk, past)
    finally:
        lib.ullsam_set_attn_variant(0)
    torch.cuda.synchronize()
    assert torch.equal(new, old), float((new.float() - old.float()).abs().max())
    # float64 reference on the operands as the kernel sees them
    qf = q.double().reshape(B, Sq, H, hd).permute(0, 2, 1, 3)
    kf = kc[:, :, :Sk].double().repeat_interleave(G, 1)
    vf = vc[:, :, :Sk].double().repeat_interleave(G, 1)
    sc = qf @ kf.transpose(-1, -2) / math.sqrt(hd)
    fmin = float(torch.finfo(torch.float32).min)
    qpos = past + torch.arange(Sq, device=DEV)[:, None]
    add = torch.where(torch.arange(Sk, device=DEV)[None, :] > qpos, fmin, 0.0)[None, None] + torch.where(mask[:, None, None, :] == 0, fmin, 0.0)
    ref = (torch.softmax((sc + add).float(), -1).double() @ vf).permute(0, 2, 1, 3).reshape(B * Sq, H * hd)
    valid = (mask[:, past:] != 0).reshape(-1)
    assert float((new.double() - ref)[valid].abs().max()) < 4e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,KVH,G,K,bias", [(2, 70, 1, 2, 256, False), (1, 1081, 2, 4, 512, True), (4, 300, 8, 4, 256, False),
                                              (4, 1081, 5, 3, 256, True)])   # 25 head slots = N 3200 = 12.5 tiles of 256: the last tile's upper half lies past N (221 tiles -> the 256x256 kernel)
def test_wqkv_gemm_with_rope_epilogue(ops, dtype, B, S, KVH, G, K, bias):
    """wqkv projection with the head split, RoPE and the KV-cache append fused into the GEMM epilogue (modeling_internlm2.py:359-388)
    against the two-kernel path (GEMM, then rope_split) and against numpy: covers the 128x128 and 256x256 kernels and the split tail."""
    hd = 128
    rng = np.random.default_rng(S + K)
    x = rng.standard_normal((B * S, K), dtype=np.float32)
    w = (rng.standard_normal((KVH * (G + 2) * hd, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32)
    bv = rng.standard_normal(w.shape[0], dtype=np.float32) * 0.2 if bias else None
    pos = (np.arange(S, dtype=np.int32)[None] + np.arange(B, dtype=np.int32)[:, None] * 3) % (S + 5)   # per-sequence offsets
    cos, sin = O.rope_tables(hd, S + 8, 1e6)
    xd, wd = T(x, dtype), T(w, dtype)
    cap, p0 = S + 6, 2
    kc = torch.zeros((B, KVH, cap, hd), dtype=dtype, device=DEV); vc = torch.zeros_like(kc)
    q = ops.gemm_qkv_rope(xd, wd, None if bv is None else T(bv), kc, vc, T(pos, torch.int32), T(cos), T(sin), B, S, KVH, G, p0)
    kc2 = torch.zeros_like(kc); vc2 = torch.zeros_like(kc)
    qkv = ops.gemm(xd, wd, None if bv is None else T(bv))
    q2 = ops.rope_split(qkv, kc2, vc2, T(pos, torch.int32), T(cos), T(sin), B, S, KVH, G, hd, p0)
    # numpy reference from the operands as the kernel sees them
    r = (xd.float().cpu().numpy() @ wd.float().cpu().numpy().T + (0 if bv is None else bv)).reshape(B, S, KVH, G + 2, hd)
    c, s_ = cos[pos][:, :, None, None, :], sin[pos][:, :, None, None, :]
    rot = r * c + O._rotate_half(r) * s_
    q_ref = rot[..., :G, :].reshape(B * S, KVH * G * hd)
    k_ref = rot[..., G, :].transpose(0, 2, 1, 3)
    v_ref = r[..., G + 1, :].transpose(0, 2, 1, 3)
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    assert err(q.float().cpu().numpy(), q_ref) < tol
    assert err(kc[:, :, p0:p0 + S].float().cpu().numpy(), k_ref) < tol and err(vc[:, :, p0:p0 + S].float().cpu().numpy(), v_ref) < tol
    assert float(kc[:, :, :p0].abs().max()) == 0 and float(kc[:, :, p0 + S:].abs().max()) == 0      # nothing outside the appended rows
    # the fused epilogue rotates the fp32 accumulators, the two-kernel path rotates values already rounded to the model dtype
    assert err(q.float().cpu().numpy(), q2.float().cpu().numpy()) < (1e-5 if dtype == torch.float32 else 4e-2)
    assert err(kc.float().cpu().numpy(), kc2.float().cpu().numpy()) < (1e-5 if dtype == torch.float32 else 4e-2)


def test_naive_and_fewkeys_attention(ops):
    rng = np.random.default_rng(0)
    B, H, hd, Sq, Sk = 2, 8, 16, 7, 300
    q = rng.standard_normal((B, Sq, H * hd), dtype=np.float32)
    k = rng.standard_normal((B, Sk, H * hd), dtype=np.float32)
    v = rng.standard_normal((B, Sk, H * hd), dtype=np.float32)

    def ref(q, k, v):
        sp = lambda x: x.reshape(x.shape[0], x.shape[1], H, hd).transpose(0, 2, 1, 3)
        a = O.softmax(np.matmul(sp(q), sp(k).transpose(0, 1, 3, 2)) / np.float32(math.sqrt(hd)))
        return np.matmul(a, sp(v)).transpose(0, 2, 1, 3).reshape(q.shape)

    st = lambda S: (S * H * hd, H * hd, hd)
    out = ops.naive_attention(T(q), T(k), T(v), B, H, H, hd, Sq, Sk, st(Sq), st(Sk), st(Sk), st(Sq), 1 / math.sqrt(hd))
    assert err(out.cpu().numpy().reshape(q.shape), ref(q, k, v)) < 1e-5
    out = ops.fewkeys_attention(T(k), T(q), T(q), B, H, hd, Sk, Sq, 1 / math.sqrt(hd))  # image -> tokens
    assert err(out.cpu().numpy().reshape(k.shape), ref(k, q, q)) < 1e-5
    out = ops.fewkeys_attention(T(k[:1]), T(q), T(q), B, H, hd, Sk, Sq, 1 / math.sqrt(hd), q_shared=True)
    assert err(out.cpu().numpy().reshape(k.shape), ref(np.repeat(k[:1], B, 0), q, q)) < 1e-5


@pytest.mark.parametrize("B,H,KVH,Sk", [(4, 32, 8, 1100), (1, 16, 8, 70), (3, 8, 1, 513), (2, 32, 8, 5)])
def test_decode_attention(ops, B, H, KVH, Sk):
    """q_len == 1 against the bf16 KV cache: GQA groups together, split-K softmax merge, left-padding mask (finfo.min additive)."""
    rng = np.random.default_rng(Sk)
    hd, cap = 128, Sk + 7
    q = T(rng.standard_normal((B, H * hd), dtype=np.float32), torch.bfloat16)
    kc = T(rng.standard_normal((B, KVH, cap, hd), dtype=np.float32), torch.bfloat16)
    vc = T(rng.standard_normal((B, KVH, cap, hd), dtype=np.float32), torch.bfloat16)
    mask = np.ones((B, Sk), np.int32)
    if Sk > 8:
        mask[0, :3] = 0                                  # left padding on the first sequence
    qf = q.float().cpu().numpy().reshape(B, H, hd)
    kf = kc.float().cpu().numpy()[:, :, :Sk]
    vf = vc.float().cpu().numpy()[:, :, :Sk]
    G = H // KVH
    sc = np.einsum("bhd,bhkd->bhk", qf, np.repeat(kf, G, 1)) / np.float32(math.sqrt(hd))
    sc = sc + np.where(mask[:, None, :] == 0, np.finfo(np.float32).min, 0).astype(np.float32)
    ref = np.einsum("bhk,bhkd->bhd", O.softmax(sc), np.repeat(vf, G, 1)).reshape(B, H * hd)
    for km in (T(mask, torch.int32), None):
        out = ops.decode_attention(q, kc, vc, km, B, H, KVH, hd, Sk).float().cpu().numpy()
        if km is None:
            sc2 = np.einsum("bhd,bhkd->bhk", qf, np.repeat(kf, G, 1)) / np.float32(math.sqrt(hd))
            ref2 = np.einsum("bhk,bhkd->bhd", O.softmax(sc2), np.repeat(vf, G, 1)).reshape(B, H * hd)
            assert err(out, ref2) < 2e-2
        else:
            assert err(out, ref) < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("P,Tq,N", [(1, 7, 4096), (3, 8, 1000), (70, 6, 4096), (2, 1, 64), (5, 7, 130), (4, 12, 777), (2, 16, 33)])
def test_tok2img_attention(ops, dtype, P, Tq, N):
    """Token -> image cross attention streamed over the keys (split-K softmax merge), per-prompt and shared K/V.  bf16 K / V run on the matrix pipe
    (tok2img_partial_mfma_kernel, up to 16 query tokens, queries and probabilities as two bf16 terms: the same 2e-5 bound as the fp32 VALU kernel)."""
    if Tq > 8 and dtype == torch.float32:
        pytest.skip("fp32 K / V: the VALU kernel, at most 8 query tokens")
    rng = np.random.default_rng(P * 1000 + Tq)
    H, hd = 8, 16
    q = rng.standard_normal((P, Tq, H * hd), dtype=np.float32) * 2
    k = rng.standard_normal((P, N, H * hd), dtype=np.float32)
    v = rng.standard_normal((P, N, H * hd), dtype=np.float32)
    kd, vd = T(k).to(dtype), T(v).to(dtype)
    kr, vr = kd.float().cpu().numpy(), vd.float().cpu().numpy()

    def ref(q, k, v):
        sp = lambda x: x.reshape(x.shape[0], x.shape[1], H, hd).transpose(0, 2, 1, 3)
        a = O.softmax(np.matmul(sp(q), sp(k).transpose(0, 1, 3, 2)) / np.float32(math.sqrt(hd)))
        return np.matmul(a, sp(v)).transpose(0, 2, 1, 3).reshape(q.shape)

    out = ops.tok2img_attention(T(q).reshape(P * Tq, -1), kd.reshape(P * N, -1), vd.reshape(P * N, -1), P, H, hd, Tq, N, 1 / math.sqrt(hd))
    assert err(out.cpu().numpy().reshape(q.shape), ref(q, kr, vr)) < 2e-5
    out = ops.tok2img_attention(T(q).reshape(P * Tq, -1), kd[:1].reshape(N, -1).contiguous(), vd[:1].reshape(N, -1).contiguous(), P, H, hd, Tq, N,
                                1 / math.sqrt(hd), kv_shared=True)
    assert err(out.cpu().numpy().reshape(q.shape), ref(q, np.repeat(kr[:1], P, 0), np.repeat(vr[:1], P, 0))) < 2e-5


def test_data_movement_kernels(ops):
    rng = np.random.default_rng(1)
    x = rng.random((2, 3, 32, 48), dtype=np.float32)
    cols = ops.patch_im2col(T(x), 64, 16, torch.float32).cpu().numpy()
    xp = np.pad(x, ((0, 0), (0, 0), (0, 32), (0, 16)))
    ref = xp.reshape(2, 3, 4, 16, 4, 16).transpose(0, 2, 4, 1, 3, 5).reshape(2 * 16, 768)
    assert err(cols, ref) == 0
    e = rng.standard_normal((2, 64, 64, 256), dtype=np.float32)
    w = rng.standard_normal(1024, dtype=np.float32); b = rng.standard_normal(1024, dtype=np.float32)
    y = ops.pixel_shuffle_ln(T(e), T(w), T(b), 2, 64, 64, 256, 1e-5, torch.float32).cpu().numpy()
    ref = O.layer_norm(O.pixel_shuffle_v2(e).reshape(2, 1024, 1024), w, b, 1e-5).reshape(2048, 1024)
    assert err(y, ref) < 2e-5
    f = rng.standard_normal((2, 1024, 1024), dtype=np.float32)
    u = ops.pixel_unshuffle(T(f), 2, 64, 64, 256).cpu().numpy()
    s = f.reshape(2, 32, 32, 1024).transpose(0, 2, 1, 3)
    s = s.reshape(2, 32, 64, 512).transpose(0, 2, 1, 3).reshape(2, 64, 64, 256)  # [n, Y, X, c]  (NHWC of :268's NCHW)
    # oracle path: text_aware_dense_feature without the MLP
    n, h, w_, c = 2, 32, 32, 1024
    g = f.reshape(n, 32, 32, c).transpose(0, 2, 1, 3)
    g = g.reshape(n, h, 64, 512).transpose(0, 2, 1, 3).reshape(n, 64, 64, 256).transpose(0, 3, 1, 2)
    assert err(u.reshape(2, 64, 64, 256).transpose(0, 3, 1, 2), g) == 0
    t = ops.transpose(T(e.reshape(2, 4096, 256)), 2, 4096, 256).cpu().numpy()
    assert err(t, e.reshape(2, 4096, 256).transpose(0, 2, 1)) == 0
    c3 = ops.im2col3x3(T(e[:, :8, :8, :64].copy()), 2, 8, 8, 64).cpu().numpy()
    ep = np.pad(e[:, :8, :8, :64], ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = np.concatenate([ep[:, ky:ky + 8, kx:kx + 8] for ky in range(3) for kx in range(3)], -1).reshape(128, 576)
    assert err(c3, ref) == 0
    low = rng.standard_normal((2, 1, 256, 256), dtype=np.float32)
    up, m = ops.resize_bilinear(T(low), (1024, 1024), threshold=0.0)
    ref = O.bilinear_resize(low, (1024, 1024))
    assert err(up.cpu().numpy(), ref) < 1e-5
    assert (m.cpu().numpy().astype(bool) != (ref > 0)).mean() < 1e-5
    lg = rng.standard_normal((3, 1000), dtype=np.float32); lg[1, 7] = lg[1, 900] = 50.0
    assert ops.argmax(T(lg)).cpu().tolist() == lg.argmax(-1).tolist()


@pytest.mark.parametrize("variant", [1, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_tile_variants_agree_with_oracle(ops, variant, dtype):
    """The 128x128 and the 256x256 two-buffer kernel (both dtypes) through every epilogue, ragged M and N."""
    from ullsam_amd import _lib
    from ullsam_amd.packing import pack_w13
    lib = _lib.load()
    rng = np.random.default_rng(variant)
    M, N, K = 777, 640, 256
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    res = rng.standard_normal((M, N), dtype=np.float32)
    ad, wd = T(a, dtype), T(w, dtype)
    ref = ad.float().cpu().numpy() @ wd.float().cpu().numpy().T
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    try:
        lib.ullsam_set_gemm_variant(variant)
        y = ops.gemm(ad, wd, bias=T(bias), act=ops.ACT_GELU, residual=T(res), out_f32=True).cpu().numpy()
        assert err(y, O.gelu(ref + bias) + res) < tol
        y = ops.gemm(ad, wd[:600].contiguous(), out_f32=False).float().cpu().numpy()  # N = 600: ragged last tile
        assert err(y, ref[:, :600]) < (tol if dtype == torch.float32 else 6e-2)
        w1, w3 = wd[:256].contiguous(), wd[256:512].contiguous()
        y = ops.gemm(ad, pack_w13(w1, w3), act=ops.ACT_SWIGLU, out_f32=True).cpu().numpy()
        assert err(y, O.silu(ref[:, :256]) * ref[:, 256:512]) < tol
    finally:
        lib.ullsam_set_gemm_variant(0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_split_k_tail_matches_unsplit(ops, dtype):
    """576 tiles on 512 slots: the 64 tail tiles are cut into 8 K-ranges (fp32 partials + reduce kernel).  Must equal the
    unsplit launch to accumulation-order noise, through bias + residual + the SwiGLU epilogue."""
    from ullsam_amd import _lib
    from ullsam_amd.packing import pack_w13
    lib = _lib.load()
    rng = np.random.default_rng(11)
    M, N, K = 2300, 4096, 2048
    a = T(rng.standard_normal((M, K), dtype=np.float32), dtype)
    w = T((rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32), dtype)
    bias, res = T(rng.standard_normal(N, dtype=np.float32)), T(rng.standard_normal((M, N), dtype=np.float32))
    try:
        lib.ullsam_set_gemm_variant(1 | 64)
        y0 = ops.gemm(a, w, bias=bias, residual=res, out_f32=True)
        s0 = ops.gemm(a, pack_w13(w[:2048].contiguous(), w[2048:].contiguous()), act=ops.ACT_SWIGLU, out_f32=True)
        lib.ullsam_set_gemm_variant(1)
        y1 = ops.gemm(a, w, bias=bias, residual=res, out_f32=True)
        s1 = ops.gemm(a, pack_w13(w[:2048].contiguous(), w[2048:].contiguous()), act=ops.ACT_SWIGLU, out_f32=True)
    finally:
        lib.ullsam_set_gemm_variant(0)
    ref = a.float().cpu().numpy() @ w.float().cpu().numpy().T + bias.cpu().numpy() + res.cpu().numpy()
    tol = 5e-4 if dtype == torch.float32 else 2e-2
    assert err(y0.cpu().numpy(), ref) < tol and err(y1.cpu().numpy(), ref) < tol
    assert err(y1.cpu().numpy(), y0.cpu().numpy()) < 1e-3 and err(s1.cpu().numpy(), s0.cpu().numpy()) < 1e-3


@pytest.mark.parametrize("M", [1, 4, 5, 8])
def test_gemm_decode_rows(ops, M):
    """M <= 8 rows in bf16 take the weight-streaming kernel (decode step): plain, bias + GELU, fp32 residual, SwiGLU, ragged N."""
    from ullsam_amd.packing import pack_w13
    rng = np.random.default_rng(M)
    for N, K in ((4096, 4096), (1003, 512), (2048, 14336 if M <= 4 else 1536), (17923, 1024), (18432, 2048)):   # the last: SwiGLU through the resident-workgroup kernel (M <= 4)
        a = T(rng.standard_normal((M, K), dtype=np.float32), torch.bfloat16)
        w = T((rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32), torch.bfloat16)
        bias, res = T(rng.standard_normal(N, dtype=np.float32)), T(rng.standard_normal((M, N), dtype=np.float32))
        af, wf = a.float().cpu().numpy(), w.float().cpu().numpy()
        ref = af @ wf.T
        assert err(ops.gemm(a, w, out_f32=True).cpu().numpy(), ref) < 2e-3
        assert err(ops.gemm(a, w, bias=bias, act=ops.ACT_GELU).float().cpu().numpy(), O.gelu(ref + bias.cpu().numpy())) < 2e-2
        y = res.clone()
        ops.gemm(a, w, bias=bias, residual=y, out_f32=True, out=y)
        assert err(y.cpu().numpy(), ref + bias.cpu().numpy() + res.cpu().numpy()) < 2e-3
        if N % 256 == 0:
            h = N // 2
            s = ops.gemm(a, pack_w13(w[:h].contiguous(), w[h:].contiguous()), act=ops.ACT_SWIGLU, out_f32=True).cpu().numpy()
            g, u = ref[:, :h], ref[:, h:]
            assert err(s, g / (1 + np.exp(-g)) * u) < 2e-3


@pytest.mark.parametrize("M", [1, 3, 4])
@pytest.mark.parametrize("K", [2048, 4096])
def test_decode_gemm_with_rmsnorm_prologue_equals_norm_then_gemm(ops, M, K):
    """ullsam_gemm_rmsnorm (the decode step's ffn_norm + w13, and a plain / biased narrow matrix) against the two separate launches:
    the staged row is the one ullsam_norm writes (same element-to-thread assignment and order of sums), so the results are equal bit for bit."""
    from ullsam_amd.packing import pack_w13
    rng = np.random.default_rng(100 * M + K)
    x = T(rng.standard_normal((M, K), dtype=np.float32) * 3)
    nw = T(1 + 0.1 * rng.standard_normal(K, dtype=np.float32))
    F = 1024 if K == 2048 else 9216    # the second: the persistent kernel (>= 4096 four-row units)
    w13 = pack_w13(T(rng.standard_normal((F, K), dtype=np.float32) / math.sqrt(K), torch.bfloat16), T(rng.standard_normal((F, K), dtype=np.float32) / math.sqrt(K), torch.bfloat16))
    wn = T(rng.standard_normal((1536, K), dtype=np.float32) / math.sqrt(K), torch.bfloat16)
    xn = ops.norm(x, nw, None, 1e-5, torch.bfloat16, rms=True)
    # oracle for the norm itself
    xf = x.cpu().numpy()
    ref = xf / np.sqrt((xf.astype(np.float64) ** 2).mean(-1, keepdims=True) + 1e-5) * nw.cpu().numpy()
    assert err(xn.float().cpu().numpy(), ref) < 4e-2     # bf16 rounding of |values| < 16
    a = ops.gemm_rmsnorm(x, nw, 1e-5, w13, act=ops.ACT_SWIGLU)
    b = ops.gemm(xn, w13, act=ops.ACT_SWIGLU)
    assert torch.equal(a, b)
    a = ops.gemm_rmsnorm(x, nw, 1e-5, wn)
    b = ops.gemm(xn, wn)
    assert torch.equal(a, b)
    if K == 4096:   # a wide plain matrix (N > 8192): the resident-workgroup kernel's general epilogue behind the norm prologue
        wb = T(rng.standard_normal((17412, K), dtype=np.float32) / math.sqrt(K), torch.bfloat16)
        a = ops.gemm_rmsnorm(x, nw, 1e-5, wb)
        ref2 = xn.float().cpu().numpy() @ wb.float().cpu().numpy().T
        assert err(a.float().cpu().numpy(), ref2) < 3e-2


@pytest.mark.parametrize("B,KVH,G,past", [(4, 8, 4, 1081), (1, 2, 4, 0), (3, 4, 2, 17)])
@pytest.mark.parametrize("normed", [False, True])
def test_decode_qkv_rope_equals_gemm_then_rope_split(ops, B, KVH, G, past, normed):
    """ullsam_decode_qkv_rope (wqkv + head split + RoPE + KV-cache append in one launch, optional RMSNorm prologue) against
    norm -> gemm -> rope_split.  The separate path rounds qkv to bf16 before rotating, the fused epilogue rotates the fp32 sums: equal to
    bf16 rounding (2^-8 relative), cache rows other than the appended one untouched, v rows equal bit for bit."""
    hd, K, cap = 128, 2048, past + 8
    rng = np.random.default_rng(B * 100 + KVH)
    N = KVH * (G + 2) * hd
    x = T(rng.standard_normal((B, K), dtype=np.float32))
    nw = T(1 + 0.1 * rng.standard_normal(K, dtype=np.float32))
    w = T(rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K), torch.bfloat16)
    bias = T(0.1 * rng.standard_normal(N, dtype=np.float32))
    pos = T(rng.integers(0, past + 1, size=(B, 1)).astype(np.int32), torch.int32)
    inv = 1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))
    fr = np.arange(past + 4, dtype=np.float32)[:, None] * inv[None]
    emb = np.concatenate([fr, fr], -1)
    cos, sin = T(np.cos(emb).astype(np.float32)), T(np.sin(emb).astype(np.float32))
    xn = ops.norm(x, nw, None, 1e-5, torch.bfloat16, rms=True) if normed else x.to(torch.bfloat16)
    fill = lambda: (torch.full((B, KVH, cap, hd), 7.0, dtype=torch.bfloat16, device="cuda"), torch.full((B, KVH, cap, hd), -3.0, dtype=torch.bfloat16, device="cuda"))
    k0, v0 = fill()
    q0 = ops.rope_split(ops.gemm(xn, w, bias), k0, v0, pos, cos, sin, B, 1, KVH, G, hd, past)
    k1, v1 = fill()
    q1 = ops.decode_qkv_rope(x if normed else xn, nw if normed else None, 1e-5, w, bias, k1, v1, pos, cos, sin, B, KVH, G, past)
    assert torch.equal(v0, v1)                      # bias + sum rounded once in both paths
    assert err(q1.float().cpu().numpy(), q0.float().cpu().numpy()) < 4e-2     # |values| < 8: three bf16 roundings apart
    assert err(k1[:, :, past].float().cpu().numpy(), k0[:, :, past].float().cpu().numpy()) < 4e-2
    untouched = torch.ones(cap, dtype=torch.bool, device="cuda"); untouched[past] = False
    assert torch.equal(k1[:, :, untouched], k0[:, :, untouched]) and bool((k1[:, :, untouched] == 7.0).all())
    # and against the exact rotation of the fp32 products
    qkv = (xn.float().cpu().numpy() @ w.float().cpu().numpy().T + bias.cpu().numpy()).reshape(B, KVH, G + 2, hd)
    c, s_ = np.cos(emb)[pos.cpu().numpy()[:, 0]][:, None, None, :], np.sin(emb)[pos.cpu().numpy()[:, 0]][:, None, None, :]
    rot = np.concatenate([-qkv[..., hd // 2:], qkv[..., :hd // 2]], -1)
    ro = qkv * c + rot * s_
    assert err(q1.float().cpu().numpy().reshape(B, KVH, G, hd), ro[:, :, :G]) < 2e-2   # one rounding of the result
    assert err(k1[:, :, past].float().cpu().numpy(), ro[:, :, G]) < 2e-2


@pytest.mark.parametrize("M,N,K", [(4324, 4096, 4096), (16384, 1280, 5120)])
def test_gemm_256_split_k_tail_matches_unsplit(ops, M, N, K):
    """256x256 kernel: 272 tiles (16-tile tail cut 8 ways, 4-row reduce slabs) and 320 tiles (64-tile tail cut 4 ways, 16-row slabs):
    the split launch + reduce kernel must equal the unsplit launch to accumulation-order noise, through bias + fp32 residual."""
    from ullsam_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(M)
    a = T(rng.standard_normal((M, K), dtype=np.float32), torch.bfloat16)
    w = T((rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32), torch.bfloat16)
    bias, res = T(rng.standard_normal(N, dtype=np.float32)), T(rng.standard_normal((M, N), dtype=np.float32))
    try:
        lib.ullsam_set_gemm_variant(3 | 64)
        y0 = ops.gemm(a, w, bias=bias, residual=res, out_f32=True)
        b0 = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU)
        lib.ullsam_set_gemm_variant(3)
        y1 = ops.gemm(a, w, bias=bias, residual=res, out_f32=True)
        b1 = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU)
    finally:
        lib.ullsam_set_gemm_variant(0)
    assert err(y1.cpu().numpy(), y0.cpu().numpy()) < 1e-3 and err(b1.float().cpu().numpy(), b0.float().cpu().numpy()) < 2e-2
    rows = rng.choice(M, 64, replace=False)
    ref = a[rows].float().cpu().numpy() @ w.float().cpu().numpy().T + bias.cpu().numpy() + res[rows].cpu().numpy()
    assert err(y1[rows].cpu().numpy(), ref) < 2e-2


def test_gemm256_split_k_tail_matches_unsplit(ops):
    """272 tiles of 256x256 on 256 CUs: the 16 tail tiles are cut into 8 K-ranges.  Equal to the unsplit launch (accumulation-order
    noise) with residual and SwiGLU epilogues; sampled rows also against numpy."""
    from ullsam_amd import _lib
    from ullsam_amd.packing import pack_w13
    lib = _lib.load()
    rng = np.random.default_rng(12)
    M, N, K = 4324, 4096, 4096
    a = T(rng.standard_normal((M, K), dtype=np.float32), torch.bfloat16)
    w = T((rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).astype(np.float32), torch.bfloat16)
    res = T(rng.standard_normal((M, N), dtype=np.float32))
    w13 = pack_w13(w[:2048].contiguous(), w[2048:].contiguous())
    try:
        lib.ullsam_set_gemm_variant(3 | 64)
        y0 = ops.gemm(a, w, residual=res, out_f32=True)
        s0 = ops.gemm(a, w13, act=ops.ACT_SWIGLU, out_f32=True)
        lib.ullsam_set_gemm_variant(3)
        for _ in range(5):
            y1 = ops.gemm(a, w, residual=res, out_f32=True)
            s1 = ops.gemm(a, w13, act=ops.ACT_SWIGLU, out_f32=True)
            assert err(y1.cpu().numpy(), y0.cpu().numpy()) < 2e-3 and err(s1.cpu().numpy(), s0.cpu().numpy()) < 2e-3
    finally:
        lib.ullsam_set_gemm_variant(0)
    rows = np.r_[0:32, 4200:4324]
    ref = a.float().cpu().numpy()[rows] @ w.float().cpu().numpy().T + res.cpu().numpy()[rows]
    assert err(y1.cpu().numpy()[rows], ref) < 2e-2


def _implied_abs_error(got_bf16: torch.Tensor, exact64: torch.Tensor):
    """A bf16 output cannot show a 1e-5 error directly (its ulp is 2^-8 relative), but it shows it statistically: an approximation error e
    moves a value across a rounding boundary with probability 2 e / ulp.  Per binade of the exact result: mismatch share x ulp / 2 = the
    implied mean absolute error of the function BEFORE rounding; also returns the largest mismatch in ulps (must be 1)."""
    want = exact64.to(torch.float32).to(torch.bfloat16)
    neq = got_bf16 != want
    mag = exact64.abs()
    worst, worst_ulps = 0.0, 0
    for e in range(-9, 3):
        sel = (mag >= 2.0 ** e) & (mag < 2.0 ** (e + 1))
        n = int(sel.sum())
        if n < 20000:
            continue
        ulp = 2.0 ** (e - 7)
        worst = max(worst, float(neq[sel].float().mean()) * ulp / 2)
        if bool(neq[sel].any()):
            worst_ulps = max(worst_ulps, int(((got_bf16[sel].float() - want[sel].float()).abs() / ulp).max().round()))
    return worst, worst_ulps


@pytest.mark.parametrize("variant,M,N", [(8, 2048, 640), (9, 2176, 512)])
def test_ring_kernel_gelu_epilogue_to_1e5_of_exact_erf(ops, variant, M, N):
    """The production bf16 ring kernels evaluate GELU with a degree-5 erfc fit (`gelu_erfc5`) that the fp32 parity mode never executes.  Drive
    that epilogue with pre-activations known exactly (one-hot operands: acc = x_m * c_n exactly, + an fp32 bias) and compare with the exact erf
    form in float64: implied absolute error < 1e-5 in every binade, never more than one bf16 ulp off."""
    from ullsam_amd import _lib
    lib = _lib.load()
    K = 128
    g = torch.Generator(device=DEV); g.manual_seed(variant)
    xm = (torch.rand(M, device=DEV, generator=g) * 12 - 6).bfloat16()              # pre-activations over [-6.5, 6.5]
    cn = torch.tensor([0.25, 0.5, 1.0], device=DEV)[torch.randint(0, 3, (N,), device=DEV, generator=g)]
    bias = (torch.rand(N, device=DEV, generator=g) - 0.5)
    a = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16); a[:, 0] = xm
    w = torch.zeros(N, K, device=DEV, dtype=torch.bfloat16); w[:, 0] = cn.bfloat16()
    try:
        lib.ullsam_set_gemm_variant(variant)
        got = ops.gemm(a, w, bias, act=ops.ACT_GELU)
        torch.cuda.synchronize()
    finally:
        lib.ullsam_set_gemm_variant(0)
    pre = (xm.float()[:, None] * cn[None, :] + bias[None, :]).double()              # the kernel's fp32 pre-activation, exactly
    exact = 0.5 * pre * (1.0 + torch.erf(pre / math.sqrt(2.0)))
    e, ulps = _implied_abs_error(got, exact)
    print(f"ring variant {variant}: GELU epilogue implied |error| {e:.2e}, worst mismatch {ulps} ulp")
    assert e < 1e-5 and ulps <= 1, (e, ulps)


def test_ring_kernel_swiglu_epilogue_to_1e5_of_exact_division(ops):
    """The 272x256 ring kernel's SwiGLU epilogue multiplies by v_rcp_f32 instead of dividing (bf16 only: the fp32 parity mode keeps the
    division).  gate = x_m * c_n and up = y_m * d_n exactly (one-hot operands): the output must be bf16(silu(gate) * up) up to the implied-error bound."""
    from ullsam_amd import _lib
    from ullsam_amd.packing import pack_w13
    lib = _lib.load()
    M, I, K = 2176, 512, 128
    g = torch.Generator(device=DEV); g.manual_seed(3)
    xm = (torch.rand(M, device=DEV, generator=g) * 16 - 8).bfloat16()
    cn = torch.tensor([0.25, 0.5, 1.0, 2.0], device=DEV)[torch.randint(0, 4, (I,), device=DEV, generator=g)]
    ym = (torch.rand(M, device=DEV, generator=g) + 1.0).bfloat16()                 # up = y_m * d_n exactly: many distinct products, so that the
    dn = torch.tensor([0.5, 1.0, 2.0], device=DEV)[torch.randint(0, 3, (I,), device=DEV, generator=g)]   # mismatch share is a probability, not a count of a few values
    a = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16); a[:, 0] = xm; a[:, 1] = ym
    w1 = torch.zeros(I, K, device=DEV, dtype=torch.bfloat16); w1[:, 0] = cn.bfloat16()
    w3 = torch.zeros(I, K, device=DEV, dtype=torch.bfloat16); w3[:, 1] = dn.bfloat16()
    try:
        lib.ullsam_set_gemm_variant(9)
        got = ops.gemm(a, pack_w13(w1, w3), act=ops.ACT_SWIGLU)
        torch.cuda.synchronize()
    finally:
        lib.ullsam_set_gemm_variant(0)
    gate = (xm.float()[:, None] * cn[None, :]).double()
    exact = gate / (1.0 + torch.exp(-gate)) * (ym.float()[:, None] * dn[None, :]).double()
    e, ulps = _implied_abs_error(got, exact)
    print(f"272x256 ring: SwiGLU epilogue implied |error| {e:.2e}, worst mismatch {ulps} ulp")
    assert e < 1e-5 and ulps <= 1, (e, ulps)


@pytest.mark.parametrize("R,C,pad,dt", [(4096, 1280, 64, torch.float32), (1081, 4096, 64, torch.float32), (1000, 257, 1, torch.float32), (1001, 130, 1, torch.bfloat16),
                                        (70, 3840, 64, torch.bfloat16), (4096, 5120, 4, torch.float32)])
def test_transpose_to_bf16_against_torch(ops, R, C, pad, dt):
    """ops.transpose_to_bf16 (csrc/vit_misc.hip: the dW = dY^T X operands of the training step and the W^T of dX): [R, C] fp32 / bf16 -> bf16 [C, R rounded up to pad] with zero
    columns behind R -- the 8-byte-store form (padded row length % 4 == 0) and the 2-byte form (odd lengths), ragged tiles."""
    g = torch.Generator(device=DEV); g.manual_seed(R + C)
    x = torch.randn(R, C, device=DEV, generator=g).to(dt)
    got = ops.transpose_to_bf16(x, pad)
    Rp = -(-R // pad) * pad
    want = torch.zeros(C, Rp, dtype=torch.bfloat16, device=DEV)
    want[:, :R] = x.t().to(torch.bfloat16)
    torch.cuda.synchronize()
    assert got.shape == (C, Rp) and torch.equal(got, want)


@pytest.mark.parametrize("R,C,pad", [(4096, 1280, 64), (1081, 4096, 64), (1000, 260, 4), (70, 3840, 64)])
def test_cast_transpose_bf16_against_torch(ops, R, C, pad):
    """ops.cast_transpose_bf16 (csrc/vit_misc.hip: one pass over an fp32 activation or gradient of the training step's bf16 Linear): the bf16 copy, the zero-padded bf16 transpose
    and the column sums (64-row blocks added in order) against torch; the two bf16 outputs must be exact, the sums within fp32 summation noise of a float64 sum."""
    g = torch.Generator(device=DEV); g.manual_seed(R + C)
    x = torch.randn(R, C, device=DEV, generator=g)
    rm, xt, cs = ops.cast_transpose_bf16(x, pad, row_major=True, colsum=True)
    Rp = -(-R // pad) * pad
    want_t = torch.zeros(C, Rp, dtype=torch.bfloat16, device=DEV)
    want_t[:, :R] = x.t().to(torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(rm, x.to(torch.bfloat16)) and torch.equal(xt, want_t) and torch.equal(xt, ops.transpose_to_bf16(x, pad))
    ref = x.double().sum(0)
    assert float((cs.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())) * (R ** 0.5)
    rm2, xt2, cs2 = ops.cast_transpose_bf16(x, pad, row_major=False, colsum=False)
    assert rm2 is None and cs2 is None and torch.equal(xt2, want_t)
