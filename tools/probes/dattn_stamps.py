"""Stamps of decode_attn_partial_kernel (side build with -DULLSAM_STAMP_DECODE, ULLSAM_HIP_LIB=...stamp.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ullsam_amd import ops, _lib
B, H, KVH, hd, Sk, cap, L = 4, 32, 8, 128, 1100, 1146, 8
q = torch.randn(B, H * hd, device="cuda", dtype=torch.bfloat16)
kc = [torch.randn(B, KVH, cap, hd, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
vc = [torch.randn(B, KVH, cap, hd, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
big = torch.empty(1 << 28, device="cuda", dtype=torch.uint8)   # flush
km = torch.ones(B, Sk, device="cuda", dtype=torch.int32)
dbg = torch.zeros(32 * 16 * 4 * 8, device="cuda", dtype=torch.int64)
_lib.call("ullsam_set_attn_debug", dbg.data_ptr())
for i in range(L): ops.decode_attention(q, kc[i], vc[i], km, B, H, KVH, hd, Sk)
for trial in range(3):
    big.fill_(trial); torch.cuda.synchronize()
    dbg.zero_()
    ops.decode_attention(q, kc[trial], vc[trial], km, B, H, KVH, hd, Sk)
    torch.cuda.synchronize()
    s = dbg.cpu().numpy().reshape(-1, 8).astype(np.int64)
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    d = np.diff(s[:, :7], axis=1)
    print(f"trial {trial}: waves {len(s)}; start skew (cycles) median {np.median(s[:, 0] - t0):.0f} max {(s[:, 0] - t0).max()}; "
          f"end max {(s[:, 6] - t0).max()}")
    print("   mean cycles: q-load %.0f | first K/V chunk %.0f | rest of loop %.0f | wave merge %.0f | barrier %.0f | epilogue %.0f" % tuple(d.mean(0)))
    print("   max  cycles: q-load %.0f | first K/V chunk %.0f | rest of loop %.0f | wave merge %.0f | barrier %.0f | epilogue %.0f" % tuple(d.max(0)))
