"""The persistent ring kernel (csrc/gemm_ring8p.h) against the one-tile-per-workgroup ring kernel on the step's multi-round GEMMs, ONE process:
bit-equality of the outputs (same MFMA order, same epilogue arithmetic), then interleaved timing rounds on cold operands.
usage: python tools/persist_ab.py [rounds]      (ullsam_set_gemm_tuning(2, 0 / 1) is the switch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ullsam_amd import ops, _lib
from ullsam_amd.packing import pack_w13

lib = _lib.load()
dev = "cuda"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
MODES = [int(m) for m in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1"])]   # 0 one-tile kernel, 1 persistent, 3 persistent with request distance 3 (experiment)
SHAPES = [  # name, M, N, K, act, bias, res
    ("llm.w13", 4324, 28672, 4096, 3, False, False), ("vit.qkv", 16384, 3840, 1280, 0, True, False), ("vit.lin1", 16384, 5120, 1280, 1, True, False),
    ("llm.wqkv.plain", 4324, 6144, 4096, 0, False, False), ("vit.lin1.plain", 16384, 5120, 1280, 0, False, False), ("llm.w13.plain", 4324, 28672, 4096, 0, False, False),
    ("b8.proj+r", 32768, 1280, 1280, 0, True, True), ("b8.wo+r", 8648, 4096, 4096, 0, False, True), ("ragged.gelu", 5000, 1280, 256, 1, True, False), ("ragged.res", 5002, 2560, 384, 0, True, True),
    ("ragged.swiglu", 4000, 8192, 128, 3, False, False),
]
only = os.environ.get("GEMM_SHAPES")
for name, M, N, K, act, hb, res in SHAPES:
    if only and name not in only.split(","):
        continue
    n_out = N // 2 if act == 3 else N
    ncopy = max(2, int(6e8 // (2 * (M * K + N * K + M * n_out))) + 1)
    As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(ncopy)]
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(ncopy)]
    bias = torch.randn(N, device=dev) if hb else None
    outs = {}
    for pz in MODES:
        lib.ullsam_set_gemm_tuning(2, pz)
        if res:
            x = torch.arange(M * N, device=dev, dtype=torch.float32).reshape(M, N) * 1e-6
            ops.gemm(As[0], Ws[0], bias, act=act, out=x, out_f32=True, residual=x)
            outs[pz] = x
        else:
            outs[pz] = ops.gemm(As[0], Ws[0], bias, act=act)
    torch.cuda.synchronize()
    ref = As[0].float() @ Ws[0].float().T
    eq = {pz: (torch.equal(outs[0], outs[pz]), int((outs[0] != outs[pz]).sum().item())) for pz in MODES[1:]}   # (modes 5 - 7 are ablations: they differ by construction)
    Cs = [torch.zeros(M, n_out, device=dev, dtype=torch.float32 if res else torch.bfloat16) for _ in range(ncopy)]
    times = {pz: [] for pz in MODES}
    for r in range(rounds):
        for pz in (MODES if r % 2 == 0 else MODES[::-1]):
            lib.ullsam_set_gemm_tuning(2, pz)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(ncopy):
                ops.gemm(As[i], Ws[i], bias, act=act, out=Cs[i], out_f32=res, residual=Cs[i] if res else None)
            e1.record()
            torch.cuda.synchronize()
            times[pz].append(e0.elapsed_time(e1) / ncopy)
    fl = 2.0 * M * N * K
    t0 = sorted(times[0])[rounds // 2]
    line = f"{name:16s} M={M:6d} N={N:6d} K={K:6d} | one-tile {t0 * 1e3:8.1f} us {fl / t0 / 1e9:7.1f} TF/s"
    for pz in MODES[1:]:
        t1 = sorted(times[pz])[rounds // 2]
        line += f" | mode {pz}: {t1 * 1e3:8.1f} us {fl / t1 / 1e9:7.1f} TF/s {100 * (t1 / t0 - 1):+5.1f} % bit-equal {eq[pz][0]} ({eq[pz][1]} differ)"
    print(line, flush=True)
    del As, Ws, Cs
lib.ullsam_set_gemm_tuning(2, 2)
