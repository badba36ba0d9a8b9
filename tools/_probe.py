import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
def run(M, N, K, act, use_bias, variant, out_f32=False, n=6):
    lib.ullsam_set_gemm_variant(variant)
    As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(n)]
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(n)]
    Cs = [torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16) for _ in range(n)]
    b = torch.randn(N, device=dev) if use_bias else None
    for i in range(n): ops.gemm(As[i], Ws[i], b, act=act, out_f32=out_f32, out=Cs[i])
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): ops.gemm(As[i], Ws[i], b, act=act, out_f32=out_f32, out=Cs[i])
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    t = sorted(ts)[2]
    rounds = ((M+255)//256)*((N+255)//256)/256
    print(f"code {variant:5d} M={M} N={N} K={K} act={act} bias={use_bias}: {t*1e3:7.1f} us  ({t*1e3/rounds:6.1f} us per 256-tile wave, {rounds:.2f} waves)", flush=True)
for K in (64, 1280):
    run(16384, 5120, K, 0, False, 3)
    run(16384, 5120, K, 0, False, 3 + (4 << 8))
    run(16384, 5120, K, 1, True, 3)
run(16384, 1280, 64, 0, False, 3 + 64)          # 320 tiles = 1.25 waves
run(16384, 256*4, 64, 0, False, 3 + 64)          # exactly 1 wave of 256 tiles
run(16384, 256*4, 64, 0, False, 3 + 64 + (4 << 8))
lib.ullsam_set_gemm_variant(0)
