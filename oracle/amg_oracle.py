"""CPU oracle for the automatic-mask-generation helpers -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of /root/reference/utils/amg.py (SAM's AMG helper functions; the reference ships no generator class,
SURVEY.md section 8 row a25).  Integer / boolean work: the HIP path must match these BIT-EXACTLY.
Pinned by tests/golden/amg.npz, captured by oracle/gen_golden.py from the reference functions themselves.
Box NMS is not in the reference (it lives in torchvision, absent here): `box_nms` restates torchvision.ops.nms
(greedy, score-descending, suppress IoU > threshold, area = (x2-x1)*(y2-y1)) and is pinned only by its own properties.
"""
from __future__ import annotations

import math
from itertools import product
from typing import Any, Dict, List, Tuple

import numpy as np


def build_point_grid(n: int) -> np.ndarray:
    """amg.py:179-186: n*n points at cell centres of the unit square, x fastest."""
    off = 1 / (2 * n)
    side = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(side[None, :], (n, 1)), np.tile(side[:, None], (1, n))], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side: int, n_layers: int, scale_per_layer: int) -> List[np.ndarray]:
    """amg.py:189-197."""
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size: Tuple[int, int], n_layers: int, overlap_ratio: float):
    """amg.py:200-234."""
    im_h, im_w = im_size
    short = min(im_h, im_w)
    boxes, layers = [[0, 0, im_w, im_h]], [0]
    for i in range(n_layers):
        n = 2 ** (i + 1)
        ov = int(overlap_ratio * short * (2 / n))
        cw = int(math.ceil((ov * (n - 1) + im_w) / n))
        ch = int(math.ceil((ov * (n - 1) + im_h) / n))
        xs = [int((cw - ov) * k) for k in range(n)]
        ys = [int((ch - ov) * k) for k in range(n)]
        for x0, y0 in product(xs, ys):
            boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
            layers.append(i + 1)
    return boxes, layers


def calculate_stability_score(masks: np.ndarray, mask_threshold: float, threshold_offset: float) -> np.ndarray:
    """amg.py:156-176: IoU of the masks thresholded at +offset / -offset (one contains the other) -> float32 [N...]."""
    inter = (masks > np.float32(mask_threshold + threshold_offset)).sum(-1).sum(-1).astype(np.int32)
    union = (masks > np.float32(mask_threshold - threshold_offset)).sum(-1).sum(-1).astype(np.int32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter.astype(np.float32) / union.astype(np.float32)).astype(np.float32)


def batched_mask_to_box(masks: np.ndarray) -> np.ndarray:
    """amg.py:303-346: XYXY boxes (inclusive max coordinates), [0,0,0,0] for empty masks; int64 [..., 4]."""
    shape = masks.shape
    h, w = shape[-2:]
    m = masks.reshape(-1, h, w).astype(bool)
    rows, cols = m.any(-1), m.any(-2)
    ys, xs = np.arange(h), np.arange(w)
    bottom = (rows * ys).max(-1)
    top = (rows * ys + h * (~rows)).min(-1)
    right = (cols * xs).max(-1)
    left = (cols * xs + w * (~cols)).min(-1)
    empty = (right < left) | (bottom < top)
    out = np.stack([left, top, right, bottom], -1) * (~empty)[:, None]
    return out.reshape(*shape[:-2], 4).astype(np.int64)


def uncrop_boxes_xyxy(boxes: np.ndarray, crop_box) -> np.ndarray:
    """amg.py:237-243."""
    x0, y0 = crop_box[0], crop_box[1]
    return boxes + np.array([x0, y0, x0, y0])


def is_box_near_crop_edge(boxes: np.ndarray, crop_box, orig_box, atol: float = 20.0) -> np.ndarray:
    """amg.py:78-88."""
    b = uncrop_boxes_xyxy(boxes, crop_box).astype(np.float32)
    near_crop = np.abs(b - np.asarray(crop_box, np.float32)[None]) <= atol
    near_img = np.abs(b - np.asarray(orig_box, np.float32)[None]) <= atol
    return np.logical_and(near_crop, ~near_img).any(1)


def mask_to_rle(masks: np.ndarray) -> List[Dict[str, Any]]:
    """amg.py:107-135: uncompressed column-major RLE ({'size': [h, w], 'counts': [...]}, counts start with the 0-run)."""
    b, h, w = masks.shape
    flat = masks.transpose(0, 2, 1).reshape(b, -1).astype(bool)
    out = []
    for i in range(b):
        ch = np.nonzero(flat[i, 1:] ^ flat[i, :-1])[0]
        idx = np.concatenate([[0], ch + 1, [h * w]])
        counts = ([] if not flat[i, 0] else [0]) + (idx[1:] - idx[:-1]).tolist()
        out.append({"size": [h, w], "counts": counts})
    return out


def rle_to_mask(rle: Dict[str, Any]) -> np.ndarray:
    """amg.py:138-149."""
    h, w = rle["size"]
    m = np.empty(h * w, dtype=bool)
    i, par = 0, False
    for c in rle["counts"]:
        m[i:i + c] = par
        i += c
        par = not par
    return m.reshape(w, h).transpose()


def area_from_rle(rle) -> int:
    """amg.py:152-153."""
    return int(sum(rle["counts"][1::2]))


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_nms(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision.ops.nms semantics: indices kept, by decreasing score (ties: lower index first)."""
    order = np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))
    b = boxes.astype(np.float32)
    area = box_area(b)
    keep, dead = [], np.zeros(len(b), bool)
    for i in order:
        if dead[i]:
            continue
        keep.append(i)
        xx1 = np.maximum(b[i, 0], b[:, 0]); yy1 = np.maximum(b[i, 1], b[:, 1])
        xx2 = np.minimum(b[i, 2], b[:, 2]); yy2 = np.minimum(b[i, 3], b[:, 3])
        inter = np.maximum(xx2 - xx1, 0).astype(np.float32) * np.maximum(yy2 - yy1, 0).astype(np.float32)
        iou = inter / (area[i] + area - inter)
        dead |= iou > np.float32(iou_threshold)
    return np.asarray(keep, np.int64)
