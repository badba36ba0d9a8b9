"""Times ullsam_up1_ln_gelu at the AMG shape (64 prompts x 4096 image tokens).  usage: python tools/probes/up1_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ullsam_amd import ops
rows = 64 * 4096
g = torch.Generator(device="cuda").manual_seed(0)
srcs = [torch.randn(rows, 256, device="cuda", generator=g).bfloat16() for _ in range(3)]
w0 = (torch.randn(256, 256, device="cuda", generator=g) * 0.06).bfloat16()
b0 = torch.randn(256, device="cuda", generator=g) * 0.1
lw, lb = 1 + 0.1 * torch.randn(64, device="cuda", generator=g), 0.1 * torch.randn(64, device="cuda", generator=g)
f = lambda i: ops.up1_ln_gelu(srcs[i % 3], w0, b0, lw, lb, 1e-6)
for i in range(5): f(i)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for i in range(12): f(i)
    e1.record(); torch.cuda.synchronize()
    print(f"up1_ln_gelu: {e0.elapsed_time(e1) / 12 * 1e3:.1f} us", flush=True)
