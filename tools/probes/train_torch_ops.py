"""Which torch (ATen) ops the training step launches beside the library's kernels: torch.profiler over one ViT-H + 7B-shaped bf16 step (tools/train_step_bench.py's setup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from ullsam_amd.training import train_step_loss
from ullsam_amd.utils.synthetic import microscopy_batch
dev, dt, P = "cuda:0", torch.bfloat16, 2
m = bench.build_model("h", "7b", dt, dev)
for n, p in m.named_parameters(): p.requires_grad_(not n.startswith("language_model."))
imgs, pts = microscopy_batch([3]); x = torch.from_numpy(imgs).to(dev)
ids = torch.from_numpy(bench.make_input_ids(20, 34, seed=1)).to(dev)
coords = torch.from_numpy(np.repeat(pts, P, 0) + np.arange(P, dtype=np.float32)[:, None, None] * 37.0).to(dev)
labels = torch.ones((P, 1), dtype=torch.int32, device=dev); gt = (torch.rand((P, 1, 1024, 1024), device=dev) > 0.5).float()
def step():
    for p in m.parameters(): p.grad = None
    loss, _, _ = train_step_loss(m, x, ids, torch.ones_like(ids), (coords, labels), gt); loss.backward()
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted((e for e in ka if e.key.startswith("aten::")), key=lambda e: -e.device_time_total)
print(f"{'aten op':34s} {'calls':>6s} {'device ms':>10s} {'cpu ms':>8s}")
for e in rows[:25]:
    print(f"{e.key:34s} {e.count:6d} {e.device_time_total / 1e3:10.2f} {e.cpu_time_total / 1e3:8.2f}")
fn = sorted((e for e in ka if e.key.endswith("Fn") or e.key.endswith("FnBackward") or "Backward" in e.key), key=lambda e: -e.device_time_total)
print()
for e in fn[:30]:
    print(f"{e.key:40s} {e.count:6d} {e.device_time_total / 1e3:10.2f}")
