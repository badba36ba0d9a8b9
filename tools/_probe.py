import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ullsam_amd import ops, _lib
lib = _lib.load(); lib.ullsam_set_gemm_variant(1)
dev = "cuda"
def run(M, N, K, act, use_bias, out_f32=False, n=6):
    As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(n)]
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(n)]
    Cs = [torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16) for _ in range(n)]
    b = torch.randn(N, device=dev) if use_bias else None
    for i in range(n): ops.gemm(As[i], Ws[i], b, act=act, out_f32=out_f32, out=Cs[i])
    ts = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): ops.gemm(As[i], Ws[i], b, act=act, out_f32=out_f32, out=Cs[i])
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    t = sorted(ts)[1]
    print(f"M={M} N={N} K={K} act={act} bias={use_bias} f32out={out_f32}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:7.1f} TF/s", flush=True)
for act, ub in [(0, False), (0, True), (2, True), (1, True)]:
    run(16384, 5120, 1280, act, ub)
run(16384, 3840, 1280, 0, True)
run(16384, 3840, 1280, 1, True)
run(16384, 5120, 1280, 0, False, True)
run(16384, 7680, 1280, 0, False)
run(16384, 2560, 1280, 0, False)
