"""Times ullsam_i2t_block at the AMG shape (64 prompts x 4096 image tokens, T tokens): layer-0 form (shared inputs) and layer-1 form (per-prompt inputs).
usage: python tools/probes/i2t_probe.py [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 7
P, N = 64, 4096
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
wq, wo = (r(128, 256) * 0.05).bfloat16(), (r(256, 128) * 0.05).bfloat16()
bq, bo, lnw, lnb = r(128) * 0.1, r(256) * 0.1, 1 + 0.1 * r(256), 0.1 * r(256)
ktok, vtok, pe = r(P * T, 128), r(P * T, 128), r(N, 256)
for shared in (True, False):
    rows = N if shared else P * N
    res = r(rows, 256); xin = (res + (pe if shared else pe.repeat(P, 1))).bfloat16()
    for want_f32, want_c, use_pe in ((True, True, True), (False, True, True), (True, False, False), (False, True, False), (False, False, False)):
        f = lambda: ops.i2t_block(xin, res, wq, bq, ktok, vtok, wo, bo, lnw, lnb, 1e-5, pe if use_pe else None, P, T, N, 0.25, shared, want_f32, want_c)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        byt = P * N * 256 * (4 * want_f32 + 2 * want_c + 2 * use_pe) + rows * 256 * 6
        us = e0.elapsed_time(e1) * 100
        print(f"shared={shared} f32={want_f32} c={want_c} pe={use_pe}: {us:.1f} us, {byt / 1e6:.0f} MB -> {byt / us / 1e6:.2f} TB/s", flush=True)
