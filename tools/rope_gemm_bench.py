"""A/B the wqkv GEMM with the RoPE epilogue (ullsam_gemm_qkv_rope) under forced kernels in one process (interleaved rounds, cold operands).
usage: python tools/rope_gemm_bench.py [rounds] [variants]     3 two-buffer 256x256 (LDS-staged epilogue), 6 / 9 / 10 ring 256 / 272 / 208 rows, 0 auto"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ullsam_amd import ops, _lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["3", "6", "9", "10", "0"])]
lib = _lib.load()
dev = "cuda"
B, S, K, KVH, G = 4, 1081, 4096, 8, 4
N = KVH * (G + 2) * 128
ncopy = 6
xs = [torch.randn(B * S, K, device=dev).bfloat16() for _ in range(ncopy)]
ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(ncopy)]
kcs = [torch.zeros(B, KVH, S, 128, device=dev, dtype=torch.bfloat16) for _ in range(ncopy)]
vcs = [torch.zeros_like(kcs[0]) for _ in range(ncopy)]
pos = torch.arange(S, device=dev, dtype=torch.int32).repeat(B)
inv = 1.0 / (1e6 ** (np.arange(0, 128, 2, dtype=np.float32) / 128))
fr = np.outer(np.arange(S + 8, dtype=np.float32), inv)
cos = torch.from_numpy(np.cos(np.concatenate([fr, fr], 1)).astype(np.float32)).to(dev)
sin = torch.from_numpy(np.sin(np.concatenate([fr, fr], 1)).astype(np.float32)).to(dev)
ref = None
for v in variants:
    lib.ullsam_set_gemm_variant(v)
    q = ops.gemm_qkv_rope(xs[0], ws[0], None, kcs[0], vcs[0], pos, cos, sin, B, S, KVH, G, 0)
    if ref is None:
        ref = (q.float().clone(), kcs[0].float().clone(), vcs[0].float().clone())
    else:
        d = max((q.float() - ref[0]).abs().max().item(), (kcs[0].float() - ref[1]).abs().max().item(), (vcs[0].float() - ref[2]).abs().max().item())
        print(f"variant {v}: max abs diff vs variant {variants[0]}: {d:.3e}")
times = {v: [] for v in variants}
for r in range(rounds):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        lib.ullsam_set_gemm_variant(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(ncopy):
            ops.gemm_qkv_rope(xs[i], ws[i], None, kcs[i], vcs[i], pos, cos, sin, B, S, KVH, G, 0)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / ncopy)
lib.ullsam_set_gemm_variant(0)
fl = 2.0 * B * S * N * K
for v in variants:
    t = sorted(times[v])[len(times[v]) // 2]
    print(f"llm.wqkv+rope M={B * S} N={N} K={K}  variant {v:2d}: {t * 1e3:7.1f} us  {fl / t / 1e9:7.1f} TF/s")
