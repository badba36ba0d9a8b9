"""Exploration for the real-size AMG test (BASELINE configs[4]): distribution of stability scores / predicted IoUs over the 12288 candidate
masks of a 64x64 grid on a 2048^2 microscopy tile (ViT-H, bench init), for a few stability offsets, bf16 and fp8, so that thresholds can be
chosen that keep a non-trivial set of masks."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bench import build_model
from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
from ullsam_amd.utils.synthetic import microscopy_tile

sam = build_model("h", "none", torch.bfloat16, "cuda:0")
img, _ = microscopy_tile(7, size=2048, n_cells=40, r_range=(90.0, 260.0))
img = torch.from_numpy(img * 255.0).cuda()
for fp8 in (False, True):
    sam.image_encoder.fp8_linears = fp8
    for off in (0.1,):
        gen = SamAutomaticMaskGenerator(sam, points_per_side=64, points_per_batch=64, pred_iou_thresh=-1e9, stability_score_thresh=-1.0,
                                        stability_score_offset=off, box_nms_thresh=2.0, output_mode="uncompressed_rle")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        recs = gen.generate(img)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = np.array([r["stability_score"] for r in recs]); pi = np.array([r["predicted_iou"] for r in recs]); ar = np.array([r["area"] for r in recs])
        for pt, stt in ((0.1, 0.85), (0.1, 0.87), (0.15, 0.85), (0.15, 0.87), (0.15, 0.88), (0.2, 0.85), (0.2, 0.87), (0.2, 0.88), (0.25, 0.87)):
            print("kept at pred_iou >", pt, "stability >=", stt, ":", int(((pi > pt) & (st >= stt)).sum()), flush=True)
        print(json.dumps({"fp8": fp8, "offset": off, "records": len(recs), "seconds": round(t1 - t0, 3),
                          "stability_pct": np.percentile(st, [1, 10, 25, 50, 75, 90, 99]).round(4).tolist(),
                          "pred_iou_pct": np.percentile(pi, [1, 10, 50, 90, 99]).round(4).tolist(),
                          "area_frac_pct": (np.percentile(ar, [1, 10, 50, 90, 99]) / 2048 ** 2).round(4).tolist()}), flush=True)
