"""Throughput of the automatic mask generator (BASELINE.json configs[4] shape: 64x64 point grid on a 2048^2 tile, SAM ViT-H;
ULLSAM_FP8=1 switches the encoder's LayerNorm-fed linears to the fp8 (e4m3) MFMA path of that config).  Synthetic microscopy tile, random-init
encoder, the STRUCTURED disc-drawing decoder of ullsam_amd.utils.synthetic.blob_decoder_init (no weights exist offline, and a random decoder's
masks are full-frame textures whose boxes are all the whole tile); thresholds of tests/test_amg_gpu.py::test_generator_real_size_vit_h_2048_tile_with_box_nms:
predicted IoU 0.90, stability 0.92 at offset 1.0, box NMS at SAM's 0.7 -- every filter removes candidates and a few hundred records survive.
usage: [ULLSAM_FP8=1] python tools/amg_bench.py [points_per_side] [tile] [vit] [stability_thresh] [stability_offset] [iters] [pred_iou_thresh] [box_nms]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import build_model
from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator

side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
vit = sys.argv[3] if len(sys.argv) > 3 else "h"
stab = float(sys.argv[4]) if len(sys.argv) > 4 else 0.92
off = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 6
piou = float(sys.argv[7]) if len(sys.argv) > 7 else 0.90
nms = float(sys.argv[8]) if len(sys.argv) > 8 else 0.7
if os.environ.get("ULLSAM_GEMM_VARIANT"):
    from ullsam_amd import _lib
    _lib.load().ullsam_set_gemm_variant(int(os.environ["ULLSAM_GEMM_VARIANT"]))
if os.environ.get("ULLSAM_ATTN_VARIANT"):                                    # 16: token -> image attention on the VALU kernel (A/B of the round-5 MFMA kernel)
    from ullsam_amd import _lib
    _lib.load().ullsam_set_attn_variant(int(os.environ["ULLSAM_ATTN_VARIANT"]))
from ullsam_amd.utils.synthetic import blob_decoder_init
sam = blob_decoder_init(build_model(vit, "none", torch.bfloat16, "cuda:0"))
fp8 = os.environ.get("ULLSAM_FP8") == "1"
sam.image_encoder.fp8_linears = fp8
gen = SamAutomaticMaskGenerator(sam, points_per_side=side, points_per_batch=int(os.environ.get("AMG_PPB", "64")), pred_iou_thresh=piou, stability_score_thresh=stab,
                                stability_score_offset=off, box_nms_thresh=nms, output_mode="uncompressed_rle")
gen.pipelined = os.environ.get("AMG_PIPE", "1") != "0"    # AMG_PIPE=0: round 5's per-batch loop (three stream synchronisations per batch), for A/B
from ullsam_amd.utils.synthetic import microscopy_tile
img = torch.from_numpy(microscopy_tile(7, size=tile, n_cells=40, r_range=(90.0 * tile / 2048, 260.0 * tile / 2048))[0] * 255).cuda()
encs, alls = [], []
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tok, _ = gen._encode(img)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    recs = gen.generate(img)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it:
        encs.append(t1 - t0); alls.append(t2 - t1)
t_enc, t_all = sorted(encs)[len(encs) // 2] * (iters - 1), sorted(alls)[len(alls) // 2] * (iters - 1)     # medians over the timed tiles (the first tile is a warm-up)
print(json.dumps({"workload": f"AMG {side}x{side} points on a {tile}^2 tile, SAM ViT-{vit.upper()}, {'fp8 (e4m3) qkv/lin1 + bf16' if fp8 else 'bf16'}, {os.environ.get('AMG_PPB', '64')} prompts/batch, multimask",
                  "seconds_per_tile": round(t_all / (iters - 1), 4), "encoder_seconds": round(t_enc / (iters - 1), 4), "prompts_per_s": round(side * side / (t_all / (iters - 1)), 1),
                  "masks_kept": len(recs), "thresholds": {"pred_iou": piou, "stability": stab, "stability_offset": off, "box_nms": nms}}))
assert len(recs) > 0, "every mask was filtered: the timed tile did no NMS / RLE work"
