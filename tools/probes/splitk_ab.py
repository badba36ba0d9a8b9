"""Launches of few ring tiles under a long sum: the ring kernel's split-K form (ullsam_set_gemm_tuning(3, 2): wherever it fits; by default only where the caller passes ULLSAM_ACT_SPLITK_OK) against the 128x128 kernel that took them before (3, 0);
same process, interleaved rounds, three rotating weight copies.   usage: python tools/probes/splitk_ab.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ullsam_amd import _lib, ops
lib = _lib.load()
DEV = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [("trn.wo", 1081, 4096, 4096, "res"), ("trn.w2", 1081, 4096, 14336, "res"), ("trn.dwqkv", 1081, 4096, 6144, "f32"), ("trn.dw1", 1081, 4096, 14336, "f32"),
          ("b1.wqkv", 1081, 6144, 4096, "bias"), ("amg.lin2", 4096, 1280, 5120, "res"), ("amg.proj", 4096, 1280, 1280, "res")]
g = torch.Generator(device=DEV); g.manual_seed(0)
for name, M, N, K, mode in SHAPES:
    a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    ws = [(torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16() for _ in range(3)]
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M, N, device=DEV, generator=g)
    def run(i):
        if mode == "res": return ops.gemm(a, ws[i], bias, residual=x, out_f32=True)
        if mode == "f32": return ops.gemm(a, ws[i], out_f32=True)
        return ops.gemm(a, ws[i], bias)
    ts = {0: [], 2: []}
    outs = {}
    for v in (0, 2):
        lib.ullsam_set_gemm_tuning(3, v)
        for _ in range(3): outs[v] = run(0).float()
    torch.cuda.synchronize()
    for r in range(rounds):
        for v in ((0, 2) if r % 2 == 0 else (2, 0)):
            lib.ullsam_set_gemm_tuning(3, v)
            for i in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(i % 3); e1.record(); torch.cuda.synchronize()
                if i: ts[v].append(e0.elapsed_time(e1) * 1e3)
    lib.ullsam_set_gemm_tuning(3, 1)
    t0, t1 = sorted(ts[0]), sorted(ts[2])
    m0, m1 = t0[len(t0) // 2], t1[len(t1) // 2]
    fl = 2.0 * M * N * K
    print(f"{name:10s} M={M:5d} N={N:5d} K={K:6d} | 128x128 kernel {m0:7.1f} us {fl / m0 / 1e6:7.1f} TF/s | ring split-K {m1:7.1f} us {fl / m1 / 1e6:7.1f} TF/s  {100 * (m1 / m0 - 1):+.1f} %"
          f"   max |diff| {float((outs[2] - outs[0]).abs().max()):.3e}", flush=True)
