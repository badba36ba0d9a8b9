"""One AMG point batch through the structured decoder with the fused token / head kernels on and off: predicted IoU and low-res logits side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import build_model
from ullsam_amd.utils.synthetic import blob_decoder_init, microscopy_tile
from ullsam_amd.modeling import transformer as TR, mask_decoder as MD
sam = blob_decoder_init(build_model("b", "none", torch.bfloat16, "cuda:0"))
img = torch.from_numpy(microscopy_tile(7, size=1024, n_cells=40)[0] * 255).cuda()
mean = sam.pixel_mean.reshape(-1).float().contiguous(); std = sam.pixel_std.reshape(-1).float().contiguous()
tok = sam.image_encoder.forward_tokens(img[None], mean, std)
pe = sam.prompt_encoder
pts = torch.rand(64, 1, 2, device="cuda") * 1024
lab = torch.ones((64, 1), dtype=torch.int32, device="cuda")
sparse = pe.sparse_tokens((pts.contiguous(), lab), None)
dense = pe.dense_tokens(64, None, None)
res = {}
for tokf, headf in ((True, True), (False, False), (True, False), (False, True)):
    TR.FUSED_TOK, MD.FUSED_HEADS = tokf, headf
    low, iou = sam.mask_decoder.predict_masks_tokens(tok, pe.dense_pe_tokens(), sparse, dense, (64, 64))
    res[(tokf, headf)] = (low.float(), iou.float())
    print(f"tok {tokf} heads {headf}: iou mean {float(iou.mean()):.5f} min {float(iou.min()):.5f} max {float(iou.max()):.5f}  > 0.9: {int((iou[:, 1:] > 0.9).sum())};  low mean {float(low.mean()):.5f} absmax {float(low.abs().max()):.3f}"
          f"  exact-equal neighbours {float((low[..., 1:] == low[..., :-1]).float().mean()):.4f}")
a, b = res[(True, True)], res[(False, False)]
print("low max diff", float((a[0] - b[0]).abs().max()), "iou max diff", float((a[1] - b[1]).abs().max()))
