"""Debug aid for the four-wave GEMM (variant 7): per-k-stage and per-sub-tile error map against the production kernel."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ullsam_amd import ops, _lib
lib = _lib.load()
torch.manual_seed(0)
M = N = 256
K = int(os.environ.get("K", "256"))
w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
full = torch.randn(M, K, device='cuda').bfloat16()
def run(a, v):
    lib.ullsam_set_gemm_variant(v); o = ops.gemm(a, w).float(); lib.ullsam_set_gemm_variant(0); torch.cuda.synchronize(); return o
for t in range(K // 32):
    a = torch.zeros_like(full); a[:, 32 * t:32 * t + 32] = full[:, 32 * t:32 * t + 32]
    r = a.float() @ w.float().T
    o = run(a, 7)
    print("stage", t, "max err", (o - r).abs().max().item(), "ref max", r.abs().max().item())
r = full.float() @ w.float().T
o = run(full, 7)
e = (o - r).abs().reshape(16, 16, 16, 16).amax(dim=(1, 3))
print((e > 0.05).int())
