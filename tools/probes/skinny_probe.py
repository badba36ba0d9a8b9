"""Decode-step weight streams in isolation: each of the four per-layer GEMMs at M = 4 over rotating weight copies (> Infinity Cache),
back to back, HIP-event timed -> us per launch and TB/s; then the four as a chain over 32 'layers'.
usage: python tools/probes/skinny_probe.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
D, F, QKV = 4096, 14336, 6144
SHAPES = {"wqkv": (QKV, D, 0, False), "wo": (D, D, 0, True), "w13": (2 * F, D, 3, False), "w2": (D, F, 0, True)}
if os.environ.get("SKINNY_ONLY"):
    SHAPES = {k: v for k, v in SHAPES.items() if k in os.environ["SKINNY_ONLY"].split(",")}
if os.environ.get("SKINNY_HEAD"):
    SHAPES["head"] = (92553, D, 0, False)
L = 32 if not os.environ.get('SKINNY_HEAD') else 4
Ws = {k: [torch.randn(n, kk, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(L)] for k, (n, kk, _, _) in SHAPES.items()}
xs = {D: torch.randn(M, D, device=dev, dtype=torch.bfloat16), F: torch.randn(M, F, device=dev, dtype=torch.bfloat16)}
res = torch.zeros(M, D, device=dev, dtype=torch.float32)


def one(name, i):
    n, kk, act, r = SHAPES[name]
    if r:
        ops.gemm(xs[kk], Ws[name][i], residual=res, out_f32=True, out=res)
    else:
        ops.gemm(xs[kk], Ws[name][i], act=act)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for name, (n, kk, act, r) in SHAPES.items():
    ms = timed(lambda: [one(name, i) for i in range(L)])
    us = ms * 1e3 / L
    print(f"{name:5s} N={n:6d} K={kk:6d}: {us:7.2f} us per launch (host-issued back to back; GPU-side durations: rocprofv3 --kernel-trace + probes/ktrace_summary.py), "
          f"{n * kk * 2 / us / 1e6:5.2f} TB/s", flush=True)
ms = timed(lambda: [[one(nm, i) for nm in SHAPES] for i in range(L)])
tot = sum(n * kk * 2 for n, kk, _, _ in SHAPES.values())
print(f"chain of {len(SHAPES)} x {L} layers: {ms * 1e3 / L:7.2f} us per layer, {tot / (ms * 1e3 / L) / 1e6:5.2f} TB/s")
