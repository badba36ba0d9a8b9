"""fp8 (e4m3) vs bf16 on the ViT's LayerNorm-fed linears, same process, random data: LayerNorm + GEMM pairs as the encoder runs them.
usage: python tools/fp8_bench.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ullsam_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = "cuda"
for name, M, N, K, act in (("vit.qkv", 16384, 3840, 1280, 0), ("vit.lin1", 16384, 5120, 1280, 1), ("vitb.qkv", 16384, 2304, 768, 0)):
    x = torch.randn(M, K, device=dev) * 2 + 0.3
    lw, lb = torch.randn(K, device=dev), torch.randn(K, device=dev) * 0.1
    w = torch.randn(N, K, device=dev) * K ** -0.5
    bias = torch.randn(N, device=dev)
    wb = w.bfloat16()
    w8, sw = ops.rows_fp8(w)

    def bf16():
        xn = ops.norm(x, lw, lb, 1e-6, torch.bfloat16)
        return ops.gemm(xn, wb, bias, act=act)

    def fp8():
        q, s = ops.rows_fp8(x, lw, lb, 1e-6)
        return ops.gemm_fp8(q, s, w8, sw, bias, act=act)

    def gemm_only_bf16(xn=ops.norm(x, lw, lb, 1e-6, torch.bfloat16)):
        return ops.gemm(xn, wb, bias, act=act)

    q0, s0 = ops.rows_fp8(x, lw, lb, 1e-6)

    def gemm_only_fp8():
        return ops.gemm_fp8(q0, s0, w8, sw, bias, act=act)

    ref = bf16().float()
    d = (fp8().float() - ref).abs()
    res = {}
    for tag, fn in (("norm+gemm bf16", bf16), ("norm+gemm fp8", fp8), ("gemm bf16", gemm_only_bf16), ("gemm fp8", gemm_only_fp8)):
        ts = []
        for r in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 8)
        res[tag] = sorted(ts)[len(ts) // 2]
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K} | " + " | ".join(f"{k}: {v * 1e3:7.1f} us" + (f" {fl / v / 1e9:6.0f} TF/s" if k.startswith("gemm") else "") for k, v in res.items())
          + f" | fp8 vs bf16 output: max |d| {d.max().item():.3f}, mean {d.mean().item():.4f} (|ref| max {ref.abs().max().item():.1f})", flush=True)
