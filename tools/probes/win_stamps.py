"""s_memtime stamps of win14r_attn_kernel<4> (every wave of every workgroup): where a workgroup's lifetime goes under full load."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ullsam_amd import ops, _lib
lib = _lib.load()
B, heads, hd = 4, 16, 80
qkv = [torch.randn(B * 4096, 3 * heads * hd, device="cuda").bfloat16() for _ in range(4)]
bias = torch.randn(3 * heads * hd, device="cuda").bfloat16()
rh, rw = (torch.randn(27, hd, device="cuda") * 0.1).bfloat16(), (torch.randn(27, hd, device="cuda") * 0.1).bfloat16()
nwg = B * heads * 25
buf = torch.zeros(nwg * 7 * 8, dtype=torch.int64, device="cuda")
for i in range(8): ops.vit_attention(qkv[i % 4], rh, rw, bias, B, heads, hd, 64, 64, 14)
torch.cuda.synchronize()
lib.ullsam_set_attn_debug(buf.data_ptr())
ops.vit_attention(qkv[0], rh, rw, bias, B, heads, hd, 64, 64, 14)
torch.cuda.synchronize()
lib.ullsam_set_attn_debug(None)
t = buf.cpu().numpy().reshape(nwg, 7, 8).astype(np.int64)
t0 = t[:, :, 0].min()
names = ["start", "requests issued", "table phase done", "own pieces + loads landed", "barrier passed", "group 0 done", "group 1 done"]
print("kernel span (cycles of the s_memtime clock, 100 MHz x ? -- compare ratios):", int(t[:, :, 6].max() - t0))
d = np.diff(t[:, :, :7], axis=2)
for i in range(6):
    x = d[:, :, i].reshape(-1)
    print(f"{names[i]:28s} -> {names[i + 1]:28s}: median {int(np.median(x)):7d}  p10 {int(np.percentile(x, 10)):7d}  p90 {int(np.percentile(x, 90)):7d}")
life = (t[:, :, 6].max(1) - t[:, :, 0].min(1))
print("workgroup lifetime: median", int(np.median(life)), "p10", int(np.percentile(life, 10)), "p90", int(np.percentile(life, 90)), " sum / 512 slots =", int(life.sum() / 512))
start = np.sort(t[:, :, 0].min(1) - t0)
print("workgroup start times: 512th", int(start[511]), "1024th", int(start[1023]), "last", int(start[-1]))
wave_end_spread = t[:, :, 6].max(1) - t[:, :, 6].min(1)
print("spread of the waves' end times inside a workgroup: median", int(np.median(wave_end_spread)), "p90", int(np.percentile(wave_end_spread, 90)))
