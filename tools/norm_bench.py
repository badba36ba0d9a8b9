"""Bandwidth of the norm kernels on the pipeline's shapes (cold operands), next to a plain fp32 -> bf16 cast of the same bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ullsam_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
def t(fn, n, reps=7):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[reps // 2] * 1e3
for name, rows, D, rms in (("llm rmsnorm", 4324, 4096, True), ("vit layernorm", 16384, 1280, False)):
    n = max(2, int(6e8 // (rows * D * 6)) + 1)
    xs = [torch.randn(rows, D, device=dev) for _ in range(n)]
    outs = [torch.empty(rows, D, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    lib.ullsam_set_norm_variant(1)
    t1 = t(lambda i: ops.norm(xs[i], w, None if rms else b, 1e-6, torch.bfloat16, rms=rms, out=outs[i]), n)
    lib.ullsam_set_norm_variant(0)
    lib.ullsam_set_norm_variant(1)
    t1 = t(lambda i: ops.norm(xs[i], w, None if rms else b, 1e-6, torch.bfloat16, rms=rms, out=outs[i]), n)
    lib.ullsam_set_norm_variant(0)
    tn = t(lambda i: ops.norm(xs[i], w, None if rms else b, 1e-6, torch.bfloat16, rms=rms, out=outs[i]), n)
    tc = t(lambda i: ops.add_cast(xs[i], None, torch.bfloat16, out=outs[i]), n)
    gb = rows * D * 6 / 1e3
    print(f"{name:14s} [{rows} x {D}] fp32 -> bf16: wave per row {t1:6.1f} us ({gb / t1 / 1e3:4.2f} TB/s)   production {tn:6.1f} us ({gb / tn / 1e3:4.2f} TB/s)   cast {tc:6.1f} us ({gb / tc / 1e3:4.2f} TB/s)")
