"""rel_h only, V = one-hot of the key's window row: out[q][d] = attention mass of query q on key row d."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
B, heads, grid, hd, W = 1, 1, 14, 80, 14
g = torch.Generator(device="cuda"); g.manual_seed(1)
t = torch.zeros(196, 3, 1, hd, device="cuda")
t[:, 0, 0] = torch.randn(196, hd, device="cuda", generator=g)
t[:, 1, 0] = 0
for k in range(196):
    t[k, 2, 0, k // 14] = 1.0
qkv = t.reshape(196, 240).bfloat16()
rh = (torch.randn(27, hd, device="cuda", generator=g) * 0.1).bfloat16()
z = torch.zeros_like(rh)
bias = torch.zeros(240, device="cuda").bfloat16()
new = ops.vit_attention(qkv, rh, z, bias, 1, 1, hd, 14, 14, 14).float().reshape(14, 14, hd)[:, :, :14]
lib.ullsam_set_attn_variant(13)
old = ops.vit_attention(qkv, rh, z, bias, 1, 1, hd, 14, 14, 14).float().reshape(14, 14, hd)[:, :, :14]
lib.ullsam_set_attn_variant(0)
q = qkv.float().reshape(196, 3, hd)[:, 0]
idx = torch.arange(14, device="cuda")
Th = torch.einsum("hwc,hkc->hwk", q.reshape(14, 14, hd), rh.float()[idx[:, None] - idx[None, :] + 13])
ref = torch.softmax(Th, -1)
torch.set_printoptions(precision=3, linewidth=200, sci_mode=False)
print("max |old - ref|", float((old - ref).abs().max()), " max |new - ref|", float((new - ref).abs().max()))
for qh in (0, 1, 7):
    print(f"qh {qh} qw 3: ref {ref[qh, 3]}\n            new {new[qh, 3]}")
# which ref row does each new row look like?
for qh in range(14):
    e = [(float((new[qh, 3] - torch.softmax(Th[qh2, 3], -1)).abs().max())) for qh2 in range(14)]
    print(qh, "best matching ref qh", min(range(14), key=lambda i: e[i]), "err", round(min(e), 4), " lg-ratio new/ref", (new[qh, 3].log2() - ref[qh, 3].log2()).round(decimals=2).tolist())
