"""Wall time of one training step (ullsam_amd.training.train_step_loss + backward) at the shapes of the models the reference ships
(SAM ViT-B + InternLM2-1.8B-shaped) or the bench's (ViT-H + 7B-shaped).  Correctness of the step is gated by tests/test_train_gpu.py; this
only times it.   usage: python tools/train_step_bench.py [b|h] [2b|7b] [instances] [fp32|bf16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from ullsam_amd.training import train_step_loss
from ullsam_amd.utils.synthetic import microscopy_batch

vit = sys.argv[1] if len(sys.argv) > 1 else "b"
llm = sys.argv[2] if len(sys.argv) > 2 else "2b"
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
dev = "cuda:0"
if os.environ.get("ULLSAM_GEMM_TUNING3"):    # A/B of the frozen linears' K-range split: 0 = off
    from ullsam_amd import _lib
    _lib.load().ullsam_set_gemm_tuning(3, int(os.environ["ULLSAM_GEMM_TUNING3"]))
m = bench.build_model(vit, llm, dt, dev)
for n, p in m.named_parameters():
    p.requires_grad_(not n.startswith("language_model."))
imgs, pts = microscopy_batch([3])
x = torch.from_numpy(imgs).to(dev)
ids = torch.from_numpy(bench.make_input_ids(20, 34, seed=1)).to(dev)
coords = torch.from_numpy(np.repeat(pts, P, 0) + np.arange(P, dtype=np.float32)[:, None, None] * 37.0).to(dev)
labels = torch.ones((P, 1), dtype=torch.int32, device=dev)
gt = (torch.rand((P, 1, 1024, 1024), device=dev) > 0.5).float()
times = []
for it in range(3):
    for p in m.parameters():
        p.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, _, _ = train_step_loss(m, x, ids, torch.ones_like(ids), (coords, labels), gt)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    times.append((t1 - t0, t2 - t1))
fw, bw = times[-1]
mode = 'bf16 model (frozen LLM on bf16 GEMMs, the rest fp32 arithmetic)' if dt == torch.bfloat16 else 'fp32'
print(json.dumps({"workload": f"train step, ViT-{vit.upper()} + InternLM2-{llm}-shaped (frozen) + decoder, {mode}, {P} instances, S = {ids.shape[1]}",
                  "forward_s": round(fw, 3), "backward_s": round(bw, 3), "loss": round(float(loss.detach()), 4),
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
