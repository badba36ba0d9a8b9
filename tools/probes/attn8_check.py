"""The 8-wave / 128-key-tile global ViT attention (production on the 64x64 grid) against the 4-wave / 64-key-tile form (attn variant 9): equality and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
B, heads, hd = 4, 16, 80
qkv = (torch.randn(B * 4096, 3 * heads * hd, device="cuda") * 0.5).bfloat16()
bias = torch.randn(3 * heads * hd, device="cuda").bfloat16()
rh, rw = (torch.randn(127, hd, device="cuda") * 0.1).bfloat16(), (torch.randn(127, hd, device="cuda") * 0.1).bfloat16()
lib.ullsam_set_attn_variant(9); a = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, 64, 64, 0).float()
lib.ullsam_set_attn_variant(0); b = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, 64, 64, 0).float()
torch.cuda.synchronize()
print("max |diff|", (a - b).abs().max().item(), "scale", a.abs().max().item())
ts = {9: [], 0: []}
for r in range(7):
    for v in (9, 0):
        lib.ullsam_set_attn_variant(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, 64, 64, 0)
        e1.record(); torch.cuda.synchronize(); ts[v].append(e0.elapsed_time(e1) / 4 * 1e3)
lib.ullsam_set_attn_variant(0)
for v in ts: print(f"variant {v}: {sorted(ts[v])[3]:.1f} us")
