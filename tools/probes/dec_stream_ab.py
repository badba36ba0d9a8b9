"""Launch times of the AMG decoder's streaming kernels at the batch shape (64 prompts x 4096 image tokens) under one or more builds of the kernel library
(tools/build_side.py), interleaved rounds in one process, event pairs, three rotating inputs; outputs compared with the first build's.
usage: python tools/probes/dec_stream_ab.py [cur,name,...] [up1,kv]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ullsam_amd import _lib, ops
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cur"]
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["up1", "kv"]
libs = {}
for n in names:
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "ullsam_amd", "lib", "libullsam_hip.so" if n == "cur" else f"libullsam_hip_{n}.so")
    libs[n] = _lib.load()
DEV = "cuda:0"
rows = 64 * 4096
g = torch.Generator(device=DEV); g.manual_seed(1)
srcs = [torch.randn(rows, 256, device=DEV, generator=g).bfloat16() for _ in range(3)]
src2 = [torch.randn(rows, 256, device=DEV, generator=g).bfloat16() for _ in range(3)]
w0 = (torch.randn(256, 256, device=DEV, generator=g) * 0.08).bfloat16()
b0 = torch.randn(256, device=DEV, generator=g) * 0.1
lw = 1.0 + 0.1 * torch.randn(64, device=DEV, generator=g); lb = 0.1 * torch.randn(64, device=DEV, generator=g)
wk = (torch.randn(128, 256, device=DEV, generator=g) * 0.08).bfloat16(); wv = (torch.randn(128, 256, device=DEV, generator=g) * 0.08).bfloat16()
bk = torch.randn(128, device=DEV, generator=g) * 0.1; bv = torch.randn(128, device=DEV, generator=g) * 0.1
OPS = {"up1": lambda i: ops.up1_ln_gelu(srcs[i], w0, b0, lw, lb, 1e-6),
       "kv": lambda i: ops.kv_proj(srcs[i], src2[i], wk, bk, wv, bv)}
for op in which:
    f = OPS[op]
    outs, ts = {}, {n: [] for n in names}
    for n in names:
        _lib._lib = libs[n]
        for _ in range(5):
            o = f(0)
        outs[n] = (torch.cat(o, 1) if isinstance(o, tuple) else o).float()
    torch.cuda.synchronize()
    for r in range(12):
        for n in (names if r % 2 == 0 else names[::-1]):
            _lib._lib = libs[n]
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(i % 3); e1.record(); torch.cuda.synchronize()
                if i: ts[n].append(e0.elapsed_time(e1) * 1e3)
    for n in names:
        t = sorted(ts[n]); d = (outs[n] - outs[names[0]]).abs()
        print(f"{op:4s} 64 prompts, lib {n:8s}: median {t[len(t) // 2]:.1f} us  min {t[0]:.1f}   vs {names[0]}: max abs diff {d.max().item():.3e}, "
              f"elements differing {(d > 0).float().mean().item():.2e}")
