"""Run the bench step a few times under one attention variant (for a rocprofv3 kernel trace of the step per variant).
usage: python tools/probes/step_variant.py <attn variant> [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ullsam_amd import _lib

lib = _lib.load()
lib.ullsam_set_attn_variant(int(sys.argv[1]))
model = bench.build_model("h", "7b", torch.bfloat16, "cuda")
inputs = bench.make_inputs(4, 1081, "cuda", True)
step = bench.mask_path_compute(model, inputs, torch.bfloat16)
with torch.no_grad():
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        step()
    torch.cuda.synchronize()
