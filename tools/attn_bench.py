"""A/B the attention kernels on the workload's shapes (ViT-H window/global, InternLM2-7B causal) in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
B, heads, hd = 4, 16, 80
qkv = [torch.randn(B * 4096, 3 * heads * hd, device=dev).bfloat16() for _ in range(4)]
bias = torch.randn(3 * heads * hd, device=dev).bfloat16()
def t(fn, n=4, reps=5):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[reps // 2] * 1e3
rh14, rw14 = (torch.randn(27, hd, device=dev) * 0.1).bfloat16(), (torch.randn(27, hd, device=dev) * 0.1).bfloat16()
rh64, rw64 = (torch.randn(127, hd, device=dev) * 0.1).bfloat16(), (torch.randn(127, hd, device=dev) * 0.1).bfloat16()
H, KVH, S = 32, 8, 1081
q = (torch.randn(B * S, H * 128, device=dev) * float(os.environ.get("ATTN_QSCALE", "1"))).bfloat16()
kc = torch.randn(B, KVH, S, 128, device=dev).bfloat16(); vc = torch.randn(B, KVH, S, 128, device=dev).bfloat16()
ops.vit_attention(qkv[0], rh64, rw64, bias, B, heads, hd, 64, 64, 0); torch.cuda.synchronize()  # clocks / caches warm
for _ in range(3): ops.vit_attention(qkv[0], rh64, rw64, bias, B, heads, hd, 64, 64, 0)
torch.cuda.synchronize()
for v in (int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])):
    lib.ullsam_set_attn_variant(v)
    w = t(lambda i: ops.vit_attention(qkv[i], rh14, rw14, bias, B, heads, hd, 64, 64, 14))
    g = t(lambda i: ops.vit_attention(qkv[i], rh64, rw64, bias, B, heads, hd, 64, 64, 0))
    print(f"variant {v}: window {w:7.1f} us ({19.67e3 / w:6.1f} TF/s)   global {g:7.1f} us ({343.6e3 / g:6.1f} TF/s)", flush=True)
    c = t(lambda i: ops.causal_attention(q, kc, vc, None, B, H, KVH, 128, S, S, 0))
    print(f"variant {v}: causal {c:7.1f} us ({B * H * S * S * 128 * 2 / c / 1e6:6.1f} TF/s causal-halved)", flush=True)
lib.ullsam_set_attn_variant(0)
