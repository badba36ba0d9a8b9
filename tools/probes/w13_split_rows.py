"""llm.w13 (4324 x 28672 x 4096, SwiGLU): 1904 tiles of 256x256 = 7.44 rounds of 256 CUs.  Does cutting the launch into 4096 rows (1792
tiles = 7 whole rounds) + the ragged 228-row remainder on the 128x128 kernel beat the single launch (whose 8th round is 44 % full)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
M, N, K = 4324, 28672, 4096
n = 4
As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(n)]
Ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16() for _ in range(n)]
Cs = [torch.zeros(M, N // 2, device="cuda", dtype=torch.bfloat16) for _ in range(n)]
def one(i, vb):
    lib.ullsam_set_gemm_variant(vb); ops.gemm(As[i], Ws[i], act=3, out=Cs[i])
def two(i, vb, vr, cut=4096):
    lib.ullsam_set_gemm_variant(vb); ops.gemm(As[i][:cut], Ws[i], act=3, out=Cs[i][:cut])
    lib.ullsam_set_gemm_variant(vr); ops.gemm(As[i][cut:], Ws[i], act=3, out=Cs[i][cut:])
cases = {"one launch, two-buffer": lambda i: one(i, 128 | 3), "one launch, four-wave": lambda i: one(i, 7),
         "4096 rows four-wave + 228 rows 128x128": lambda i: two(i, 7, 1), "4096 rows four-wave + 228 rows two-buffer": lambda i: two(i, 7, 3),
         "4096 rows two-buffer + 228 rows 128x128": lambda i: two(i, 3, 1)}
ref = None
for name, fn in cases.items():
    fn(0); torch.cuda.synchronize()
    if ref is None: ref = Cs[0].float().clone()
    else: assert (Cs[0].float() - ref).abs().max().item() < 0.1, name
ts = {k: [] for k in cases}
for r in range(7):
    for name, fn in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize(); ts[name].append(e0.elapsed_time(e1) / n * 1e3)
lib.ullsam_set_gemm_variant(0)
for k, v in ts.items(): print(f"{k:50s} {sorted(v)[3]:8.1f} us")
