"""Mask fill of candidate tiles for the full-depth parity fixture (tests/golden/full_depth.npz): the bench configuration (ViT-H x 32 + 7B-shaped
InternLM2 x 32, the fixture's own weights) in the library's fp32 mode -- pinned to the reference within 5e-6 by test_full_depth_golden -- on
microscopy tiles of several seeds.  Picks tiles whose fp32 mask fill lies in 0.3 - 0.7 (where IoU is least forgiving) for oracle/gen_golden.py.
usage: python tools/probes/fill_scan.py [first_seed] [last_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from ullsam_amd import ops
from ullsam_amd.utils.synthetic import fill_model_like_fixtures, microscopy_batch

a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 26)
dev = "cuda"
m = fill_model_like_fixtures(bench.build_model("h", "7b", torch.float32, dev, init=False), 0)
ids = torch.from_numpy(bench.make_input_ids(20, 34, seed=1, batch=1)).to(dev)
with torch.no_grad():
    for seed in range(a, b + 1):
        x_np, pts_np = microscopy_batch([seed])
        inputs = (torch.from_numpy(x_np).to(dev), torch.from_numpy(pts_np).to(dev), torch.ones((1, 1), dtype=torch.int32, device=dev), ids)
        low, mk = bench.mask_path_compute(m, inputs, torch.float32)()
        lo = low.float().reshape(-1)
        print(f"seed {seed:3d}: fp32 mask fill {mk.float().mean().item():.4f}  low-res logits mean {lo.mean().item():+.3f} std {lo.std().item():.3f} "
              f"share |x| < 0.0085: {(lo.abs() < 0.0085).float().mean().item():.4f}", flush=True)
