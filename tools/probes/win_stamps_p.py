"""s_memtime stamps of win14p_attn_kernel<4> (third problem of every wave): where a problem's time goes in the persistent kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ullsam_amd import ops, _lib
lib = _lib.load()
B, heads, hd = 4, 16, 80
qkv = [torch.randn(B * 4096, 3 * heads * hd, device="cuda").bfloat16() for _ in range(4)]
bias = torch.randn(3 * heads * hd, device="cuda").bfloat16()
rh, rw = (torch.randn(27, hd, device="cuda") * 0.1).bfloat16(), (torch.randn(27, hd, device="cuda") * 0.1).bfloat16()
nwg = 256
buf = torch.zeros(nwg * 14 * 8, dtype=torch.int64, device="cuda")
for i in range(8): ops.vit_attention(qkv[i % 4], rh, rw, bias, B, heads, hd, 64, 64, 14)
torch.cuda.synchronize()
lib.ullsam_set_attn_debug(buf.data_ptr())
ops.vit_attention(qkv[0], rh, rw, bias, B, heads, hd, 64, 64, 14)
torch.cuda.synchronize()
lib.ullsam_set_attn_debug(None)
t = buf.cpu().numpy().reshape(nwg, 14, 8).astype(np.int64)
names = ["top", "own pieces landed", "barrier passed", "flush done", "requests + decode done", "table phase + QK done", "softmax done", "PV + pack done (next top)"]
ok = (t > 0).all(2)
print("waves with all 8 stamps:", int(ok.sum()), "of", ok.size)
for i in range(7):
    x = (t[:, :, i + 1] - t[:, :, i])[ok]
    print(f"{names[i]:26s} -> {names[i + 1]:28s}: median {int(np.median(x)):7d}  p10 {int(np.percentile(x, 10)):7d}  p90 {int(np.percentile(x, 90)):7d}")
x = (t[:, :, 7] - t[:, :, 0])[ok]
print("whole problem: median", int(np.median(x)), "p10", int(np.percentile(x, 10)), "p90", int(np.percentile(x, 90)))
