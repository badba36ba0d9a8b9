"""Shader-clock stamps of the PERSISTENT ring kernel (csrc/gemm_ring8p.h, ullsam_set_gemm_variant bits 15 + 16): per workgroup, wave group and tile (first 8 of a workgroup)
  [0] tile start (first load slot), [1] K loop done, [2] epilogue issued (conversions + stores), [3] border wait done (vmcnt(0): stages 1' / 2' landed, stores acknowledged)
-> cycles per stage inside the loop, epilogue, border wait, and the tile-to-tile period, medians over workgroups; the one-tile kernel's stamps beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ullsam_amd import ops, _lib
lib = _lib.load()
SH = {"w13": (9, 4324, 28672, 4096, 3, 272, 256, 68), "w13plain": (9, 4324, 28672, 4096, 0, 272, 256, 68), "lin1": (8, 16384, 5120, 1280, 1, 256, 320, 80), "lin1plain": (8, 16384, 5120, 1280, 0, 256, 320, 80),
      "qkv": (8, 16384, 3840, 1280, 0, 256, 320, 80)}
for name in (sys.argv[1:] or ["w13", "qkv", "lin1"]):
    var, M, N, K, act, BM, BN, mf = SH[name]
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda") if act in (0, 1) and name in ("lin1", "qkv") else None
    tiles = ((M + BM - 1) // BM) * (N // BN)
    ns = K // 32
    for mode, bits, tune in (("persistent (default: uniform trips, epilogues together)", 3 << 15, 2), ("persistent, first form", 3 << 15, 1), ("one-barrier", 3 << 15, 4), ("one-tile", 1 << 15, 2)):
        lib.ullsam_set_gemm_tuning(2, tune)
        lib.ullsam_set_gemm_variant(var | bits)
        ws = ops._gemm_workspace(a.device); ws[48 << 20:56 << 20].zero_()
        for _ in range(10): ops.gemm(a, w, bias, act=act)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(a, w, bias, act=act)
        e1.record(); torch.cuda.synchronize()
        lib.ullsam_set_gemm_variant(0)
        lib.ullsam_set_gemm_tuning(2, 2)
        raw = ws[48 << 20:56 << 20].view(torch.int64).cpu().numpy().astype(np.int64)
        us = e0.elapsed_time(e1) * 100
        if mode == "one-tile":
            st = raw[: tiles * 8].reshape(tiles, 2, 4)
            loop = np.median(st[:, :, 1] - st[:, :, 0]); epi = np.median(st[:, :, 2] - st[:, :, 1]); pro = np.median(st[:, 0, 0] - st[:, 0, 3])
            print(f"{name} one-tile:   {tiles} tiles {BM}x{BN}; prologue {pro:.0f}, loop {loop:.0f} = {loop / ns:.0f} cycles/stage ({16 * mf} matrix cycles), epilogue {epi:.0f}; sum {pro + loop + epi:.0f} per tile; launch {us:.0f} us")
        else:
            G = min(tiles, 256)
            st = raw[: G * 2 * 8 * 4].reshape(G, 2, 8, 4)
            per = (tiles + G - 1) // G
            nt = min(8, tiles // G)   # tiles every workgroup has
            loop = np.median(st[:, :, :nt, 1] - st[:, :, :nt, 0]); epi = np.median(st[:, :, :nt, 2] - st[:, :, :nt, 1]); drain = np.median(st[:, :, :nt, 3] - st[:, :, :nt, 2])
            period = np.median(st[:, :, 1:nt, 0] - st[:, :, :nt - 1, 0]) if nt > 1 else float("nan")
            span = np.median(st[:, :, nt - 1, 3] - st[:, :, 0, 0])
            print(f"{name} {mode}: {G} workgroups x {per} tiles; loop {loop:.0f} = {loop / ns:.0f} cycles/stage, epilogue {epi:.0f}, border wait {drain:.0f}; tile period {period:.0f}; first start -> last end {span:.0f} over {nt} tiles; launch {us:.0f} us")
            ins = raw[1 << 19:(1 << 19) + G * 64].reshape(G, 8, 8)[:, :, :6]
            ok = ins[:, :, 0] > 0
            if ok.any():
                d = ins - ins[:, :, :1]
                for wv in range(8):
                    m = np.median(d[ok[:, wv], wv, :], axis=0)
                    print(f"     in-stage stamps (tile 1, trip 4, stage 1; cycles from the slot's start) wave {wv}: " + " ".join(f"[{k}] {m[k]:.0f}" for k in range(1, 6)))
                g0, g1 = np.median(ins[:, 0, 0][ok[:, 0]] - ins[:, 4, 0][ok[:, 4]]), 0
                print(f"     wave 0 slot start minus wave 4 slot start: {g0:.0f} cycles")
            for g in (0, 1):
                l = np.median(st[:, g, :nt, 1] - st[:, g, :nt, 0]); e = np.median(st[:, g, :nt, 2] - st[:, g, :nt, 1]); d = np.median(st[:, g, :nt, 3] - st[:, g, :nt, 2])
                print(f"     wave group {g}: loop {l:.0f}, epilogue {e:.0f}, border wait {d:.0f}")
