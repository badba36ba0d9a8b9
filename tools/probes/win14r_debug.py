"""Where win14r_attn_kernel differs from win14_attn_kernel (variant 13): max |d| per (image, head, window) and per window row / column."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
B, heads, grid = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 2, 16)))
hd, W, D = 80, 14, heads * 80
g = torch.Generator(device="cuda"); g.manual_seed(1)
qkv = torch.randn(B * grid * grid, 3 * D, device="cuda", generator=g).bfloat16()
rh = (torch.randn(27, hd, device="cuda", generator=g) * 0.1).bfloat16()
rw = (torch.randn(27, hd, device="cuda", generator=g) * 0.1).bfloat16()
bias = (torch.randn(3 * D, device="cuda", generator=g) * 0.3).bfloat16()
new = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, W)
lib.ullsam_set_attn_variant(13)
old = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, W)
lib.ullsam_set_attn_variant(0)
d = (new.float() - old.float()).abs().reshape(B, grid, grid, heads, hd)
print("max diff", float(d.max()), "nan", int(torch.isnan(new.float()).sum()))
nw = (grid + W - 1) // W
for b in range(B):
    for h in range(heads):
        for wy in range(nw):
            for wx in range(nw):
                blk = d[b, wy * W:(wy + 1) * W, wx * W:(wx + 1) * W, h]
                print(f"b{b} h{h} w({wy},{wx}) max {float(blk.max()):.4f}  rows {[round(float(x), 3) for x in blk.amax((1, 2))]}  cols {[round(float(x), 3) for x in blk.amax((0, 2))]} dims16 {[round(float(x),3) for x in blk.amax((0,1)).reshape(5,16).amax(1)]}")

def run(tag, qkv, rh, rw, bias):
    new = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, W)
    lib.ullsam_set_attn_variant(13)
    old = ops.vit_attention(qkv, rh, rw, bias, B, heads, hd, grid, grid, W)
    lib.ullsam_set_attn_variant(0)
    d = (new.float() - old.float()).abs().reshape(B, grid, grid, heads, hd)
    print(f"{tag}: max diff {float(d.max()):.4f}; window(0,0) head 0: {float(d[0, :14, :14, 0].max()):.4f}")
    return new, old

z = torch.zeros_like(rh)
run("no rel-pos", qkv, z, z, bias)
run("rel_h only", qkv, rh, z, bias)
run("rel_w only", qkv, z, rw, bias)
q2 = qkv.clone().reshape(-1, 3, heads, hd); q2[:, 0] = 0; q2 = q2.reshape(qkv.shape)
n, o = run("q = 0 (uniform attention)", q2, rh, rw, bias)
q3 = qkv.clone().reshape(-1, 3, heads, hd); q3[:, 2] = 1.0; q3 = q3.reshape(qkv.shape)
b1 = bias.clone().reshape(3, heads, hd); b1[2] = 1.0; b1 = b1.reshape(-1)
n, o = run("v = 1", q3, rh, rw, b1)
print(" new sample", n.float().reshape(B, grid, grid, heads, hd)[0, 0, :3, 0, :4], " old", o.float().reshape(B, grid, grid, heads, hd)[0, 0, :3, 0, :4])
