"""The windowed attention launch alone (ViT-H shape, batch 4) under a list of attention variants: for rocprofv3 --pmc runs and quick A/B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
B, heads, hd = 4, 16, 80
qkv = [torch.randn(B * 4096, 3 * heads * hd, device="cuda").bfloat16() for _ in range(4)]
bias = torch.randn(3 * heads * hd, device="cuda").bfloat16()
rh, rw = (torch.randn(27, hd, device="cuda") * 0.1).bfloat16(), (torch.randn(27, hd, device="cuda") * 0.1).bfloat16()
reps = int(os.environ.get("REPS", "20"))
for v in (int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])):
    lib.ullsam_set_attn_variant(v)
    for i in range(8): ops.vit_attention(qkv[i % 4], rh, rw, bias, B, heads, hd, 64, 64, 14)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): ops.vit_attention(qkv[i % 4], rh, rw, bias, B, heads, hd, 64, 64, 14)
    e1.record(); torch.cuda.synchronize()
    print(f"variant {v}: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us", flush=True)
lib.ullsam_set_attn_variant(0)
