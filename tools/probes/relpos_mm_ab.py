"""The three products of the decomposed relative-position bias (training.BmmNTFn: forward C = A B^T, dA = dC B, dB = dC^T A) at the windowed and global shapes of ViT-H,
timed through training._mm (its choice of kernel and of the k split).   usage: python tools/probes/relpos_mm_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ullsam_amd import _lib, training as T
lib = _lib.load()
DEV = "cuda:0"
g = torch.Generator(device=DEV); g.manual_seed(0)
for name, nb, M, N, K in (("windowed", 14, 25 * 14 * 16, 14, 80), ("global", 64, 64 * 16, 64, 80)):
    A = torch.randn(nb, M, K, device=DEV, generator=g); B = torch.randn(nb, N, K, device=DEV, generator=g); dC = torch.randn(nb, M, N, device=DEV, generator=g)
    C = torch.empty(nb, M, N, device=DEV); dA = torch.empty_like(A); dB = torch.empty_like(B)
    prods = {"fwd": lambda: T._mm(A, B, C, M, N, K, (M * K, K, 1), (N * K, 1, K), (M * N, N, 1), batch=nb),
             "dA": lambda: T._mm(dC, B, dA, M, K, N, (M * N, N, 1), (N * K, K, 1), (M * K, K, 1), batch=nb),
             "dB": lambda: T._mm(dC, A, dB, N, K, M, (M * N, 1, N), (M * K, K, 1), (N * K, K, 1), batch=nb)}
    for pn, f in prods.items():
        res = {}
        for mode in (1,):
            for _ in range(3): f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
            res.setdefault(mode, []).extend(ts)
            res[("o", mode)] = {"fwd": C, "dA": dA, "dB": dB}[pn].clone()
        m = {k: sorted(res[k])[len(res[k]) // 2] for k in (1,)}
        print(f"{name:8s} {pn:3s}: {m[1]:7.1f} us", flush=True)
