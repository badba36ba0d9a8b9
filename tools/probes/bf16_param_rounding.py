"""Where does the bf16 mode's stage error exceed the reference's autocast error?  Hypothesis: a model that
IS bfloat16 (`model.to(bfloat16)`, what app.py / the trainer run) holds its norm weights and biases rounded to bf16, a per-channel SYSTEMATIC error
that the reference's torch.autocast run (fp32 parameters, bf16 matmul operands only) does not have.  Experiment: run tile 0 of
tests/golden/full_depth.npz in bf16 mode (a) as is, (b) with the fp32 copies of every 1-D parameter (norm weights, biases: the PackCache's ':f32'
entries) replaced by the unrounded fp32 values, and print the stage errors next to the reference's autocast error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from tests import test_model_gpu as TM
from tests import util as U

g = U.gold("full_depth")
m32 = TM._fill_model_from_rule(bench.build_model("h", "7b", torch.float32, "cuda", init=False), 0)
sd32 = {k: v.clone() for k, v in m32.state_dict().items() if v.dim() == 1 or v.numel() == 1}
mb = bench.build_model("h", "7b", torch.bfloat16, "cuda", init=False)
mb.load_state_dict({k: v.to(torch.bfloat16) for k, v in m32.state_dict().items()}, strict=False)
del m32
torch.cuda.empty_cache()
a = TM._full_depth_run(mb, g, torch.bfloat16, 0)
# inject unrounded fp32 values for the ':f32' packs
n = 0
for mod_name, mod in mb.named_modules():
    packs = getattr(mod, "_packs", None)
    if packs is None:
        continue
    for pname, p in mod.named_parameters(recurse=False):
        full = f"{mod_name}.{pname}" if mod_name else pname
        if full in sd32:
            for key in list(packs._store):
                if key.endswith(":f32"):
                    k0, val = packs._store[key]
                    if val.shape == sd32[full].shape and torch.allclose(val, sd32[full], rtol=1e-2, atol=1e-2):
                        packs._store[key] = (k0, sd32[full].float().contiguous()); n += 1
print(f"replaced {n} fp32 packs by unrounded values")
b = TM._full_depth_run(mb, g, torch.bfloat16, 0)
print("stage        bf16 model: mean|d|   with fp32 norm weights / biases   (reference autocast)")
for k in TM.FULL_KEYS:
    ref = g[f"{k}_0"].astype(np.float64)
    print(f"{k:11s} {np.abs(a[k] - ref).mean():10.5f}   {np.abs(b[k] - ref).mean():10.5f}   ({float(g[f'{k}_0_ac_mean_err']):.5f})")
