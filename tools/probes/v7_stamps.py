"""Shader-clock stamps of the four-wave GEMM (variant 7 | bit 15): cycles per 32-deep stage and per epilogue, median over workgroups,
for the full loop and for compile-time ablations of it (results of the ablated builds are garbage; timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ullsam_amd import ops, _lib
SH = {"qkv": (16384, 3840, 1280, 0, False), "lin1": (16384, 5120, 1280, 1, True), "w13": (4324, 28672, 4096, 3, False), "wqkv": (4324, 6144, 4096, 0, False)}
lib = _lib.load()
for name in (sys.argv[1:] or ["w13", "qkv"]):
    M, N, K, act, hb = SH[name]
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda") if hb else None
    for abl, tag in ((0, "full"), (1, "no DMA requests"), (6, "requests from wave 0 only"), (7, "MFMAs only"), (8, "no epilogue")):
        lib.ullsam_set_gemm_variant(7 | 32768 | 64 | (abl << 8))
        ws = ops._gemm_workspace(a.device); ws[48 << 20:56 << 20].zero_()
        for _ in range(20): ops.gemm(a, w, bias, act=act)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, w, bias, act=act)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        lib.ullsam_set_gemm_variant(0)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        st = ws[48 << 20:56 << 20].view(torch.int64).cpu().numpy()[: tiles * 8].reshape(tiles, 8).astype(np.int64)
        ns = K // 32
        pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], np.maximum(st[:, 3] - st[:, 2], 0)
        print(f"{name} [{tag}]: {tiles} tiles; prologue {np.median(pro):.0f}  loop {np.median(loop):.0f} = {np.median(loop) / ns:.0f} cycles/stage (64 MFMAs = 1024)  "
              f"epilogue {np.median(epi):.0f}  total {np.median(st[:, 3] - st[:, 0]):.0f} cycles; shader clock {np.median((st[:, 3] - st[:, 0]) / np.maximum(st[:, 5] - st[:, 4], 1)) * 0.1:.2f} GHz; launch {us:.0f} us")
