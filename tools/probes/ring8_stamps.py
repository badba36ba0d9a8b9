"""Shader-clock stamps of the ring kernels (variants 8 / 9 with bit 15): cycles per 32-deep stage and per epilogue, median over workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ullsam_amd import ops, _lib
from ullsam_amd.packing import pack_w13
lib = _lib.load()
SH = {"w13": (9, 4324, 28672, 4096, 3, 272, 256, 36 + 32), "w13plain": (9, 4324, 28672, 4096, 0, 272, 256, 68), "lin1": (8, 16384, 5120, 1280, 1, 256, 320, 80), "lin1plain": (8, 16384, 5120, 1280, 0, 256, 320, 80), "wo": (9, 4324, 4096, 4096, 0, 272, 256, 68), "w2": (9, 4324, 4096, 14336, 0, 272, 256, 68),
      "qkv": (8, 16384, 3840, 1280, 0, 256, 320, 80), "lin2": (8, 16384, 1280, 5120, 0, 256, 320, 80)}
for name in (sys.argv[1:] or ["w13", "w2", "qkv", "lin2"]):
    var, M, N, K, act, BM, BN, mf = SH[name]
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    lib.ullsam_set_gemm_variant(var | 32768)
    ws = ops._gemm_workspace(a.device); ws[48 << 20:56 << 20].zero_()
    for _ in range(10): ops.gemm(a, w, act=act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(a, w, act=act)
    e1.record(); torch.cuda.synchronize()
    lib.ullsam_set_gemm_variant(0)
    tiles = ((M + BM - 1) // BM) * ((N + BN - 1) // BN)
    st = ws[48 << 20:56 << 20].view(torch.int64).cpu().numpy()[: tiles * 8].reshape(tiles, 2, 4).astype(np.int64)
    ns = K // 32
    loop = np.median(st[:, :, 1] - st[:, :, 0]); epi = np.median(st[:, :, 2] - st[:, :, 1])
    print(f"{name}: {tiles} tiles of {BM}x{BN}; loop {loop:.0f} = {loop / ns:.0f} cycles/stage ({mf} MFMAs per SIMD and stage = {16 * mf} matrix cycles)  epilogue {epi:.0f}  launch {e0.elapsed_time(e1) * 100:.0f} us")
    pro = np.median(st[:, 0, 0] - st[:, 0, 3])   # first instruction of the workgroup -> loop entry (setup, two stages requested, the first landed)
    per_wg = pro + loop + epi
    rounds = (tiles + 255) // 256
    print(f"   prologue {pro:.0f} cycles; prologue + loop + epilogue = {per_wg:.0f} cycles per workgroup, x {rounds} rounds = {per_wg * rounds:.0f}")
