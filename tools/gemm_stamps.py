"""Where a unit boundary of the persistent GEMM spends its time: s_memtime stamps written by gemm256p_kernel in its diagnostic mode
(variant bit 15; lane 0 of waves 0 and 4 = the two wave groups, first four units of every workgroup).
usage: python tools/gemm_stamps.py [shape ...]     shapes: qkv lin1 w13 wqkv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ullsam_amd import ops, _lib

SH = {"qkv": (16384, 3840, 1280, 0, False), "lin1": (16384, 5120, 1280, 1, True), "w13": (4324, 28672, 4096, 3, False), "wqkv": (4324, 6144, 4096, 0, False)}
lib = _lib.load()
dev = "cuda"
for name in (sys.argv[1:] or ["qkv", "lin1"]):
    M, N, K, act, has_bias = SH[name]
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev) if has_bias else None
    for extra, tag in ((0, "full epilogue"), (8 << 8, "no stores"), (4 << 8, "no epilogue")):
        lib.ullsam_set_gemm_variant(4 | 32768 | extra)
        ws = ops._gemm_workspace(a.device)
        ws[48 << 20:56 << 20].zero_()
        for _ in range(3):
            ops.gemm(a, w, bias, act=act)
        torch.cuda.synchronize()
        st = ws[48 << 20:56 << 20].view(torch.int64).cpu().numpy().astype(np.int64)
        # the same launch timed from outside: 8 back-to-back launches between two events, stamped and unstamped builds of the call
        outs = {}
        for code, lab in ((4 | 32768 | extra, "stamped"), (4 | extra, "unstamped"), (128 | extra, "non-persistent")):
            lib.ullsam_set_gemm_variant(code)
            ops.gemm(a, w, bias, act=act)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm(a, w, bias, act=act)
            e1.record()
            torch.cuda.synchronize()
            outs[lab] = e0.elapsed_time(e1) / 8 * 1e3
        print("    event-timed per launch (8 back to back, same operands): " + ", ".join(f"{k} {v:.1f} us" for k, v in outs.items()))
        lib.ullsam_set_gemm_variant(0)
        G = 256
        st = st[: G * 2 * 4 * 16].reshape(G, 2, 4, 16)   # [workgroup][group][unit][stamp]
        nk = K // 64
        rows = []
        for grp in (0, 1):
            u0, u1 = st[:, grp, 1], st[:, grp, 2]          # boundary between unit 1 and unit 2 of every workgroup
            ok = (u0[:, 0] > 0) & (u1[:, 9] > 0)
            d = lambda x, y: np.median((x - y)[ok]) / 1e3  # kilo-cycles
            rows.append((grp, int(ok.sum()), d(u0[:, 1], u0[:, 0]), d(u0[:, 2], u0[:, 1]), d(u1[:, 3], u0[:, 2]), d(u1[:, 4], u1[:, 3]), d(u1[:, 5], u1[:, 4]),
                         d(u1[:, 6], u1[:, 5]), d(u1[:, 7], u1[:, 6]), d(u1[:, 8], u1[:, 7]), d(u1[:, 9], u1[:, 8]),
                         np.median(((u1[:, 10] - u1[:, 9]) / max(nk - 5, 1))[ok]) / 1e3, d(u1[:, 0], u0[:, 0])))
        for grp in (0, 1):
            u1 = st[:, grp, 2]
            ok = (u1[:, 7] > 0) & (u1[:, 15] > 0)
            md = lambda x, y: np.median((x - y)[ok]) / 1e3
            print(f"    step 3 of a unit, group {grp}: L0 DMA issue {md(u1[:, 11], u1[:, 7]):.2f}, fragment-read issue {md(u1[:, 15], u1[:, 11]):.2f}, "
                  f"barrier + C0 + L1 reads + DMA wait {md(u1[:, 8], u1[:, 15]):.2f}, barrier + C1 + barrier {md(u1[:, 9], u1[:, 8]):.2f} kilo-cycles")
        print(f"{name} {tag}: kilo-cycles (median over workgroups), boundary between a workgroup's 2nd and 3rd unit")
        print("  grp  n   dma-issue  epilogue  ->L0-barrier  step1:C0+L1wait  C1   step2:L0..L1wait  C1   step3:L0..L1wait  C1   steady/step  unit-period")
        for r in rows:
            print("  %d  %3d   %7.2f   %7.2f   %9.2f   %13.2f  %5.2f  %13.2f  %5.2f  %13.2f  %5.2f  %9.2f   %9.1f" % r)
        t0, t1, nu = st[:, 0, 0, 12], st[:, 0, 0, 13], st[:, 0, 0, 14]
        base = t0.min()
        fin = (t1 - base) / 100.0          # s_memrealtime: 100 MHz -> microseconds
        beg = (t0 - base) / 100.0
        print(f"    wall clock (s_memrealtime): workgroup start p50 {np.median(beg):.1f} max {beg.max():.1f} us; finish min {fin.min():.1f} p50 {np.median(fin):.1f} p95 {np.percentile(fin, 95):.1f} max {fin.max():.1f} us; "
              f"units per workgroup {int(nu.min())}..{int(nu.max())}; finish of 4-unit workgroups p50 {np.median(fin[nu == nu.max()]):.1f}, of the others {np.median(fin[nu < nu.max()]) if (nu < nu.max()).any() else float('nan'):.1f}")
