"""Automatic mask generation on top of the HIP-backed Sam: point grid -> batched prompt encoder + mask decoder ->
IoU / stability filtering -> boxes -> NMS -> RLE.

The reference ships only the helper functions (utils/amg.py) -- no generator class exists in it (SURVEY.md section 0 / 8(f)
row 2), so the driver below follows the flow those helpers were written for (Meta's SamAutomaticMaskGenerator: per crop, per
batch of `points_per_batch` single-point prompts with multimask output; keep masks with predicted IoU > pred_iou_thresh and
stability >= stability_score_thresh; drop boxes touching an interior crop edge; box NMS per crop, then across crops preferring
small crops).  Helper parity is pinned (tests/golden/amg.npz); driver parity is checked against a numpy re-statement of the same
flow over the oracle (tests/test_amg_gpu.py).

Deviations, stated: images are resized with the bilinear kernel (align_corners=False, no antialias) instead of PIL's antialiased
uint8 resize; `min_mask_region_area` post-processing labels regions with scipy instead of OpenCV (utils.amg.remove_small_regions);
the predicted-IoU filter is applied before
the masks are upsampled (it depends only on the IoU head, so the surviving set is identical, and the 3x1024^2-per-prompt logits of
rejected masks are never materialised).  With `fused_postprocess=True` (default) the upsample, stability score, mask->box and RLE
steps of a batch run as one kernel over the low-res logits (`utils.amg.postprocess_low_res`): the full-resolution fp32 logits and
uint8 masks (16 + 4 MiB per mask on a 2048^2 tile) are never written; `fused_postprocess=False` calls the reference's helpers
one after the other.  Both produce the same records.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch

from . import ops
from .utils import amg as A


class SamAutomaticMaskGenerator:
    def __init__(self, model, points_per_side: Optional[int] = 32, points_per_batch: int = 64, pred_iou_thresh: float = 0.88,
                 stability_score_thresh: float = 0.95, stability_score_offset: float = 1.0, box_nms_thresh: float = 0.7,
                 crop_n_layers: int = 0, crop_nms_thresh: float = 0.7, crop_overlap_ratio: float = 512 / 1500,
                 crop_n_points_downscale_factor: int = 1, point_grids: Optional[List[np.ndarray]] = None,
                 min_mask_region_area: int = 0, output_mode: str = "binary_mask", fused_postprocess: bool = True) -> None:
        assert (points_per_side is None) != (point_grids is None), "Exactly one of points_per_side or point_grid must be provided."
        self.point_grids = (A.build_all_layer_point_grids(points_per_side, crop_n_layers, crop_n_points_downscale_factor)
                            if points_per_side is not None else point_grids)
        assert output_mode in ("binary_mask", "uncompressed_rle", "coco_rle"), f"Unknown output_mode {output_mode}."
        self.min_mask_region_area = int(min_mask_region_area)
        self.model = model
        self.points_per_batch = points_per_batch
        self.pred_iou_thresh = pred_iou_thresh
        self.stability_score_thresh = stability_score_thresh
        self.stability_score_offset = stability_score_offset
        self.box_nms_thresh = box_nms_thresh
        self.crop_n_layers = crop_n_layers
        self.crop_nms_thresh = crop_nms_thresh
        self.crop_overlap_ratio = crop_overlap_ratio
        self.output_mode = output_mode
        self.fused_postprocess = fused_postprocess

    # -- image side --------------------------------------------------------------------------------------------------
    def _encode(self, crop: torch.Tensor):
        """crop fp32 [3,h,w] in 0..255 -> (image tokens [1,N,C], input_size (h', w') in the 1024 frame)."""
        sam = self.model
        S = sam.image_encoder.img_size
        h, w = crop.shape[-2:]
        scale = S / max(h, w)
        nh, nw = int(h * scale + 0.5), int(w * scale + 0.5)  # ResizeLongestSide.get_preprocess_shape (utils/transforms.py:93-102)
        x = crop if (nh, nw) == (h, w) else ops.resize_bilinear(crop.contiguous(), (nh, nw))[0]
        mean = sam.pixel_mean.reshape(-1).float().contiguous()
        std = sam.pixel_std.reshape(-1).float().contiguous()
        return sam.image_encoder.forward_tokens(x[None].contiguous(), mean, std), (nh, nw)

    # -- one batch of point prompts ------------------------------------------------------------------------------------
    def _process_batch(self, points: np.ndarray, img_tok, input_size, crop_box, orig_size, image_cache=None) -> A.MaskData:
        sam = self.model
        dev = img_tok.device
        S = sam.image_encoder.img_size
        g = S // sam.image_encoder.patch_size
        ch, cw = crop_box[3] - crop_box[1], crop_box[2] - crop_box[0]
        n = points.shape[0]
        scaled_h = (points.astype(np.float64) * np.array([input_size[1] / cw, input_size[0] / ch])).astype(np.float32)   # ResizeLongestSide.apply_coords, on the host: one upload
        scaled = torch.from_numpy(np.ascontiguousarray(scaled_h[:, None, :])).to(dev)
        labels = self._ones_labels(n, dev)
        pe = sam.prompt_encoder
        sparse = pe.sparse_tokens((scaled, labels), None)
        dense = pe.dense_tokens(n, None, None)
        k = sam.mask_decoder.num_mask_tokens - 1
        low, iou = sam.mask_decoder.predict_masks_tokens(img_tok, pe.dense_pe_tokens(), sparse, dense, (g, g), image_cache=image_cache,
                                                         mask_range=(1, 1 + k))         # multimask_output=True (mask_decoder.py:100-105): masks 1 .. 3 only are ever produced
        if self.fused_postprocess:
            return self._finish_batch_fused(low.reshape(n * k, low.shape[-2], low.shape[-1]), iou, points, k, input_size, crop_box, orig_size)
        pts = torch.from_numpy(points).to(dev).float()
        flat_iou = iou.reshape(-1)
        keep = flat_iou > self.pred_iou_thresh
        idx = torch.nonzero(keep).reshape(-1)
        data = A.MaskData(iou_preds=flat_iou[idx], points=pts.repeat_interleave(k, dim=0)[idx])
        if idx.numel() == 0:
            data["boxes"] = torch.zeros((0, 4), dtype=torch.int64, device=dev)
            data["rles"] = []
            data["stability_score"] = flat_iou[idx]
            return data
        kept_low = low.reshape(n * k, low.shape[-2], low.shape[-1])[idx].contiguous()
        masks = sam.postprocess_masks(kept_low[:, None], input_size, (ch, cw))[:, 0]          # logits at crop resolution
        stab = A.calculate_stability_score(masks, sam.mask_threshold, self.stability_score_offset)
        data["stability_score"] = stab
        k2 = stab >= self.stability_score_thresh
        data.filter(k2)
        masks = masks[k2]
        binm = A.threshold_masks(masks, sam.mask_threshold)
        boxes = A.batched_mask_to_box(binm) if binm.shape[0] else torch.zeros((0, 4), dtype=torch.int64, device=dev)
        data["boxes"] = boxes
        orig_h, orig_w = orig_size
        k3 = ~A.is_box_near_crop_edge(boxes, crop_box, [0, 0, orig_w, orig_h]) if boxes.shape[0] else torch.zeros((0,), dtype=torch.bool, device=dev)
        if not bool(torch.all(k3)):
            data.filter(k3)
            binm = binm[k3]
        data["rles"] = A.mask_to_rle_pytorch(A.uncrop_masks(binm, crop_box, orig_h, orig_w))
        return data

    def _finish_batch_fused(self, low, iou, points, k, input_size, crop_box, orig_size) -> A.MaskData:
        """The filters of `_process_batch` on host copies of the per-mask scalars, after one fused device pass over all maps."""
        sam = self.model
        dev = low.device
        pp = A.postprocess_low_res(low, sam.image_encoder.img_size, input_size, crop_box, orig_size, sam.mask_threshold,
                                   self.stability_score_offset)
        iou_h = iou.float().cpu().numpy().reshape(-1)
        keep = (iou_h > np.float32(self.pred_iou_thresh)) & (pp.stability_score >= np.float32(self.stability_score_thresh))
        sel = np.nonzero(keep)[0]
        boxes = pp.boxes[sel]
        if len(sel):
            orig_h, orig_w = orig_size
            b = (boxes + np.asarray([crop_box[0], crop_box[1], crop_box[0], crop_box[1]])).astype(np.float32)
            near_crop = np.abs(b - np.asarray(crop_box, np.float32)[None, :]) <= 20.0                      # is_box_near_crop_edge
            near_image = np.abs(b - np.asarray([0, 0, orig_w, orig_h], np.float32)[None, :]) <= 20.0
            ok = ~np.logical_and(near_crop, ~near_image).any(axis=1)
            sel, boxes = sel[ok], boxes[ok]
        # the batch's survivors stay HOST arrays (a few records): they are moved to the device once per crop, for the box NMS (per batch that was ~10 small launches and copies)
        data = A.MaskData(iou_preds=iou_h[sel].astype(np.float32), points=np.repeat(points.astype(np.float32), k, axis=0)[sel],
                          stability_score=pp.stability_score[sel].astype(np.float32), boxes=boxes.reshape(-1, 4).astype(np.int64))
        data["rles"] = pp.rles(sel, as_list=False) if len(sel) else []
        return data

    # -- the fused path's batches as a three-deep software pipeline ----------------------------------------------------------
    def _run_batches_pipelined(self, pts: np.ndarray, img_tok, input_size, crop_box, orig_size, image_cache) -> A.MaskData:
        """Every batch of `_process_batch` + `_finish_batch_fused`, in the same order and with the same arithmetic, but with the host never waiting on the
        stream between batches: while batch i's prompt encoder / mask decoder / post-processing launches are queued, batch i - 1's per-mask scalars (copied back
        asynchronously into pinned buffers) are filtered on the host and its RLE emission is launched, and batch i - 2's change positions (copied back the same
        way) are turned into run lengths.  The per-batch loop of round 5 synchronised three times per batch (scalars, predicted IoUs, positions) and uploaded
        four small arrays from pageable memory: at 64 prompts per batch a quarter of a 2048^2 tile's time was the GPU waiting for the host."""
        sam = self.model
        dev = img_tok.device
        S = sam.image_encoder.img_size
        g = S // sam.image_encoder.patch_size
        ch, cw = crop_box[3] - crop_box[1], crop_box[2] - crop_box[0]
        k = sam.mask_decoder.num_mask_tokens - 1
        pe = sam.prompt_encoder
        # ResizeLongestSide.apply_coords for ALL points of the crop on the host, one upload
        scaled_all = torch.from_numpy(np.ascontiguousarray((pts.astype(np.float64) * np.array([input_size[1] / cw, input_size[0] / ch])).astype(np.float32)[:, None, :])).to(dev)
        starts = list(range(0, pts.shape[0], self.points_per_batch))
        slots = getattr(self, "_pipe_slots", None)
        if slots is None:
            slots = self._pipe_slots = [dict() for _ in range(3)]
        data = A.MaskData()
        issued, selected = {}, {}

        def issue(i):
            i0 = starts[i]
            p = pts[i0:i0 + self.points_per_batch]
            n = p.shape[0]
            sparse = pe.sparse_tokens((scaled_all[i0:i0 + n], self._ones_labels(n, dev)), None)
            dense = pe.dense_tokens(n, None, None)
            low, iou = sam.mask_decoder.predict_masks_tokens(img_tok, pe.dense_pe_tokens(), sparse, dense, (g, g), image_cache=image_cache, mask_range=(1, 1 + k))
            words, sc, frame = A.postprocess_low_res(low.reshape(n * k, low.shape[-2], low.shape[-1]), S, input_size, crop_box, orig_size, sam.mask_threshold,
                                                     self.stability_score_offset, defer=True)
            st = slots[i % 3]
            sc_h = A._pinned(st, "sc", sc.numel(), torch.int32)[:sc.numel()].view(sc.shape)
            iou_h = A._pinned(st, "iou", n * k, torch.float32)[:n * k]
            sc_h.copy_(sc, non_blocking=True)
            iou_h.copy_(iou.float().reshape(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            issued[i] = (p, words, frame, sc_h, iou_h, ev)

        def select(i):
            p, words, frame, sc_h, iou_h, ev = issued.pop(i)
            ev.synchronize()
            pp = A.PostprocessedMasks(words, sc_h.numpy(), frame)
            iou_np = iou_h.numpy().copy()
            keep = (iou_np > np.float32(self.pred_iou_thresh)) & (pp.stability_score >= np.float32(self.stability_score_thresh))
            sel = np.nonzero(keep)[0]
            boxes = pp.boxes[sel]
            if len(sel):
                orig_h, orig_w = orig_size
                b = (boxes + np.asarray([crop_box[0], crop_box[1], crop_box[0], crop_box[1]])).astype(np.float32)
                near_crop = np.abs(b - np.asarray(crop_box, np.float32)[None, :]) <= 20.0                      # is_box_near_crop_edge
                near_image = np.abs(b - np.asarray([0, 0, orig_w, orig_h], np.float32)[None, :]) <= 20.0
                ok = ~np.logical_and(near_crop, ~near_image).any(axis=1)
                sel, boxes = sel[ok], boxes[ok]
            rec = A.MaskData(iou_preds=iou_np[sel].astype(np.float32), points=np.repeat(p.astype(np.float32), k, axis=0)[sel],
                             stability_score=pp.stability_score[sel].astype(np.float32), boxes=boxes.reshape(-1, 4).astype(np.int64))
            emit = None
            if len(sel):
                emit = (*A._rle_emit_launch(words, sel, pp.rle_counts[sel], frame[0], frame[1], slots[i % 3]), pp.first[sel].copy(), frame)
            selected[i] = (rec, emit)

        def finalize(i):
            rec, emit = selected.pop(i)
            if emit is None:
                rec["rles"] = []
            else:
                pos_h, offs, ev, first, frame = emit
                ev.synchronize()
                rec["rles"] = A._rle_records(pos_h.numpy().astype(np.int64), offs, first, frame[0], frame[1], as_list=False)
            data.cat(rec, deep=False)

        nb = len(starts)
        for i in range(nb + 2):
            if i < nb:
                issue(i)
            if 0 <= i - 1 < nb:
                select(i - 1)
            if 0 <= i - 2 < nb:
                finalize(i - 2)
        return data

    def _ones_labels(self, n: int, dev) -> torch.Tensor:
        c = getattr(self, "_labels_cache", None)
        if c is None or c.shape[0] != n or c.device != torch.device(dev):
            c = torch.ones((n, 1), dtype=torch.int32, device=dev)
            self._labels_cache = c
        return c

    def postprocess_small_regions(self, data: A.MaskData, min_area: int, nms_thresh: float) -> A.MaskData:
        """Remove small islands / fill small holes of every mask, then re-run box NMS preferring masks that needed no change
        (the generator step the reference's remove_small_regions helper, utils/amg.py:267-291, exists for)."""
        if len(data["rles"]) == 0:
            return data
        dev = data["boxes"].device
        new_masks, scores = [], []
        for rle in data["rles"]:
            mask = A.rle_to_mask(rle)
            mask, changed = A.remove_small_regions(mask, min_area, mode="holes")
            unchanged = not changed
            mask, changed = A.remove_small_regions(mask, min_area, mode="islands")
            unchanged = unchanged and not changed
            new_masks.append(torch.from_numpy(np.ascontiguousarray(mask)).unsqueeze(0))
            scores.append(float(unchanged))
        masks = torch.cat(new_masks, 0).to(dev)
        boxes = A.batched_mask_to_box(masks)
        keep = A.batched_nms(boxes.float(), torch.as_tensor(scores, device=dev), torch.zeros_like(boxes[:, 0]), nms_thresh)
        for i in keep.tolist():
            if scores[i] == 0.0:
                data["rles"][i] = A.mask_to_rle_pytorch(masks[i:i + 1])[0]
                data["boxes"][i] = boxes[i]
        data.filter(keep)
        return data

    def _process_crop(self, image: torch.Tensor, crop_box, layer_idx: int, orig_size) -> A.MaskData:
        x0, y0, x1, y1 = crop_box
        img_tok, input_size = self._encode(image[:, y0:y1, x0:x1])
        pts = self.point_grids[layer_idx] * np.array([[x1 - x0, y1 - y0]], dtype=np.float64)
        data = A.MaskData()
        image_cache = {} if getattr(self, "reuse_image_side", True) else None      # what the decoder computes from the crop's embedding alone (keys = embedding + dense prompt, their model-dtype copies, layer 0's K / V): once per crop, not per point batch
        if self.fused_postprocess and getattr(self, "pipelined", True):
            data = self._run_batches_pipelined(pts, img_tok, input_size, crop_box, orig_size, image_cache)
        else:
            for (p,) in A.batch_iterator(self.points_per_batch, pts):
                data.cat(self._process_batch(p, img_tok, input_size, crop_box, orig_size, image_cache), deep=False)
        dev = img_tok.device
        for key in ("iou_preds", "points", "stability_score", "boxes"):      # the fused path collects host arrays: to the device once per crop
            if key in data._stats and isinstance(data[key], np.ndarray):
                data[key] = torch.from_numpy(data[key]).to(dev)
        if len(data["rles"]):
            keep = A.batched_nms(data["boxes"].float(), data["iou_preds"], torch.zeros_like(data["boxes"][:, 0]), self.box_nms_thresh)
            data.filter(keep)
        data["boxes"] = A.uncrop_boxes_xyxy(data["boxes"], crop_box)
        data["points"] = A.uncrop_points(data["points"], crop_box)
        data["crop_boxes"] = torch.tensor([crop_box for _ in range(len(data["rles"]))], device=data["boxes"].device).reshape(-1, 4)
        return data

    def generate_batch(self, images, group=None) -> List[List[Dict[str, Any]]]:
        """Several tiles (BASELINE configs[4]: 2048^2 tiles over the 8 GPUs of a node): tiles are independent units, so with torch.distributed
        initialised (one process per GPU) every rank generates the masks of its contiguous share of `images` (parallel.shard_range) and the
        ragged record lists are exchanged once at the end (parallel.gather_sharded_lists); every rank returns all tiles' records in tile order.
        Without a process group it is a loop over `generate`."""
        from . import parallel
        rank, ws = parallel.world(group)       # rank and size IN the group the records are gathered over
        a, b = parallel.shard_range(len(images), rank, ws)
        mine = [self.generate(images[i]) for i in range(a, b)]
        return parallel.gather_sharded_lists(mine, len(images), group)

    @torch.no_grad()
    def generate(self, image) -> List[Dict[str, Any]]:
        """image: HxWx3 uint8 / float array (0..255) or a [3,H,W] tensor.  Returns SAM-style records sorted as generated."""
        dev = self.model.device
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1)
        image = image.to(dev).float().contiguous()
        orig_size = tuple(int(v) for v in image.shape[-2:])
        crop_boxes, layer_idxs = A.generate_crop_boxes(orig_size, self.crop_n_layers, self.crop_overlap_ratio)
        data = A.MaskData()
        for crop_box, layer_idx in zip(crop_boxes, layer_idxs):
            data.cat(self._process_crop(image, crop_box, layer_idx, orig_size), deep=False)
        if len(crop_boxes) > 1 and len(data["rles"]):
            cb = data["crop_boxes"].float()
            scores = 1.0 / ((cb[:, 2] - cb[:, 0]) * (cb[:, 3] - cb[:, 1]))  # prefer masks from smaller crops
            keep = A.batched_nms(data["boxes"].float(), scores, torch.zeros_like(data["boxes"][:, 0]), self.crop_nms_thresh)
            data.filter(keep)
        if self.min_mask_region_area > 0:
            if not isinstance(data["rles"], list):
                data["rles"] = list(data["rles"])
            data = self.postprocess_small_regions(data, self.min_mask_region_area, max(self.box_nms_thresh, self.crop_nms_thresh))
        data.to_numpy()
        out = []
        for i, rle in enumerate(data["rles"]):
            if not isinstance(rle["counts"], list):
                rle = {"size": rle["size"], "counts": rle["counts"].tolist()}
            seg = A.rle_to_mask(rle) if self.output_mode == "binary_mask" else (A.coco_encode_rle(rle) if self.output_mode == "coco_rle" else rle)
            out.append({"segmentation": seg, "area": A.area_from_rle(rle), "bbox": A.box_xyxy_to_xywh(data["boxes"][i]).tolist(),
                        "predicted_iou": float(data["iou_preds"][i]), "point_coords": [data["points"][i].tolist()],
                        "stability_score": float(data["stability_score"][i]), "crop_box": A.box_xyxy_to_xywh(data["crop_boxes"][i]).tolist()})
        return out
