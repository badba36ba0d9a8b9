"""Where a wave of causal128_attn_kernel spends its cycles (stamped diagnostic build): LDS-DMA request issue, the two blocks of compute,
the wait + barrier at the end of a tile -- medians per query-block weight, the kernel's time, and the timeline of the workgroups (s_memrealtime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ullsam_amd import ops, _lib
lib = _lib.load()
B, H, KVH, S = 4, 32, 8, 1081
q = torch.randn(B * S, H * 128, device="cuda").bfloat16()
kc = torch.randn(B, KVH, S, 128, device="cuda").bfloat16(); vc = torch.randn(B, KVH, S, 128, device="cuda").bfloat16()
nwg = H * B * ((S + 127) // 128)
buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device="cuda")
for _ in range(5): ops.causal_attention(q, kc, vc, None, B, H, KVH, 128, S, S, 0)
lib.ullsam_set_attn_debug(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.causal_attention(q, kc, vc, None, B, H, KVH, 128, S, S, 0); e1.record(); torch.cuda.synchronize()
lib.ullsam_set_attn_debug(None)
st = buf.cpu().numpy().reshape(nwg, 4, 8)
print(f"stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us")
for qb in sorted(set(st[:, 0, 5])):
    sel = st[st[:, 0, 5] == qb]
    nt = sel[0, 0, 4]
    d, c, bw, tot = (np.median(sel[:, :, k]) for k in range(4))
    print(f"query block {qb}: {nt} tiles; per tile and wave: request issue {d / nt:7.0f}  compute {c / nt:7.0f}  wait+barrier {bw / nt:7.0f}  (loop {tot / nt:7.0f} cycles/tile, {tot:9.0f} total)")
t0 = st[:, :, 6].min()
beg, end = (st[:, 0, 6] - t0) / 100.0, (st[:, 0, 7] - t0) / 100.0     # us
print(f"loop entry of the first / last workgroup: {beg.min():.1f} / {beg.max():.1f} us; last loop exit {end.max():.1f} us")
for qb in sorted(set(st[:, 0, 5])):
    sel = st[:, 0, 5] == qb
    print(f"query block {qb}: loops start {np.percentile(beg[sel], [0, 50, 100]).round(1)} us, end {np.percentile(end[sel], [0, 50, 100]).round(1)} us, duration median {np.median(end[sel] - beg[sel]):.1f} us")
busy = np.zeros(int(end.max()) + 2)
for a, b in zip(beg, end):
    busy[int(a):int(b) + 1] += 1
print("workgroups inside their loop, per 5 us:", [int(busy[i:i + 5].mean()) for i in range(0, len(busy), 5)])
