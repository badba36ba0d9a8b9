"""Automatic-mask-generation helpers with the reference's names (utils/amg.py), device work on HIP kernels.

Device-side functions (stability score, mask -> box, RLE encoding, box NMS) take CUDA tensors and launch kernels from
libullsam_hip.so; they are bit-exact with the reference functions (integer / boolean arithmetic).  Pure bookkeeping
(point grids, crop boxes, MaskData, uncrop offsets) is host Python, as in the reference.
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import Any, Dict, Iterator, List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from ..ops import _chk, _stream


class MaskData:
    """Batched per-mask records: a dict of equal-length lists / arrays / tensors with filter() and cat() (amg.py:16-75)."""

    _OK = (list, np.ndarray, torch.Tensor)

    def __init__(self, **fields) -> None:
        self._stats: Dict[str, Any] = {}
        for k, v in fields.items():
            self[k] = v

    def __setitem__(self, key: str, item: Any) -> None:
        assert isinstance(item, self._OK), "MaskData only supports list, numpy arrays, and torch tensors."
        self._stats[key] = item

    def __getitem__(self, key: str) -> Any:
        return self._stats[key]

    def __delitem__(self, key: str) -> None:
        del self._stats[key]

    def items(self):
        return self._stats.items()

    def filter(self, keep: torch.Tensor) -> None:
        keep_np = keep.detach().cpu().numpy()
        for k, v in list(self._stats.items()):
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                self._stats[k] = v[torch.as_tensor(keep, device=v.device)]
            elif isinstance(v, np.ndarray):
                self._stats[k] = v[keep_np]
            elif isinstance(v, list):
                idx = np.nonzero(keep_np)[0] if keep.dtype == torch.bool else keep_np
                self._stats[k] = [v[int(i)] for i in idx]
            else:
                raise TypeError(f"MaskData key {k} has an unsupported type {type(v)}.")

    def cat(self, other: "MaskData", deep: bool = True) -> None:
        """`deep=False` skips the reference's element-wise deepcopy of list fields (the generator never mutates a batch after
        concatenating it; deep-copying 12288 RLE records per tile costs more than the GPU work)."""
        for k, v in other.items():
            cur = self._stats.get(k)
            if isinstance(v, list) and not deep:
                self._stats[k] = list(v) if cur is None else cur + v
            elif cur is None:
                self._stats[k] = deepcopy(v)
            elif isinstance(v, torch.Tensor):
                self._stats[k] = torch.cat([cur, v], dim=0)
            elif isinstance(v, np.ndarray):
                self._stats[k] = np.concatenate([cur, v], axis=0)
            elif isinstance(v, list):
                self._stats[k] = cur + deepcopy(v)
            else:
                raise TypeError(f"MaskData key {k} has an unsupported type {type(v)}.")

    def to_numpy(self) -> None:
        for k, v in self._stats.items():
            if isinstance(v, torch.Tensor):
                self._stats[k] = v.detach().cpu().numpy()


# ---- host bookkeeping ---------------------------------------------------------------------------------------------
def build_point_grid(n_per_side: int) -> np.ndarray:
    """n*n cell-centre points of the unit square, x fastest (amg.py:179-186)."""
    centres = (np.arange(n_per_side, dtype=np.float64) + 0.5) / n_per_side
    centres = np.linspace(centres[0], centres[-1], n_per_side)  # same rounding as the reference's linspace
    gx, gy = np.meshgrid(centres, centres)
    return np.stack([gx, gy], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side: int, n_layers: int, scale_per_layer: int) -> List[np.ndarray]:
    return [build_point_grid(int(n_per_side / (scale_per_layer ** layer))) for layer in range(n_layers + 1)]


def generate_crop_boxes(im_size: Tuple[int, ...], n_layers: int, overlap_ratio: float) -> Tuple[List[List[int]], List[int]]:
    """Layer i has (2**i)**2 overlapping XYXY crops; layer 0 is the whole image (amg.py:200-234)."""
    im_h, im_w = im_size
    short_side = min(im_h, im_w)
    boxes, layer_of = [[0, 0, im_w, im_h]], [0]
    for layer in range(1, n_layers + 1):
        per_side = 2 ** layer
        overlap = int(overlap_ratio * short_side * (2 / per_side))
        crop_w = int(math.ceil((overlap * (per_side - 1) + im_w) / per_side))
        crop_h = int(math.ceil((overlap * (per_side - 1) + im_h) / per_side))
        for ix in range(per_side):  # x outer, y inner: itertools.product(x0s, y0s) order
            x0 = int((crop_w - overlap) * ix)
            for iy in range(per_side):
                y0 = int((crop_h - overlap) * iy)
                boxes.append([x0, y0, min(x0 + crop_w, im_w), min(y0 + crop_h, im_h)])
                layer_of.append(layer)
    return boxes, layer_of


def batch_iterator(batch_size: int, *args) -> Iterator[List[Any]]:
    n = len(args[0])
    assert len(args) > 0 and all(len(a) == n for a in args), "Batched iteration must have inputs of all the same size."
    for start in range(0, n, batch_size):
        yield [a[start:start + batch_size] for a in args]


def box_xyxy_to_xywh(box_xyxy):
    out = deepcopy(box_xyxy)
    out[2] = out[2] - out[0]
    out[3] = out[3] - out[1]
    return out


def _offset(t: torch.Tensor, vals: Sequence[int]) -> torch.Tensor:
    off = torch.tensor([list(vals)], device=t.device)
    return off.unsqueeze(1) if t.dim() == 3 else off


def uncrop_boxes_xyxy(boxes: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    return boxes + _offset(boxes, (crop_box[0], crop_box[1], crop_box[0], crop_box[1]))


def uncrop_points(points: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    return points + _offset(points, (crop_box[0], crop_box[1]))


def uncrop_masks(masks: torch.Tensor, crop_box: List[int], orig_h: int, orig_w: int) -> torch.Tensor:
    x0, y0, x1, y1 = crop_box
    if (x0, y0, x1, y1) == (0, 0, orig_w, orig_h):
        return masks
    out = torch.zeros(masks.shape[:-2] + (orig_h, orig_w), dtype=masks.dtype, device=masks.device)
    out[..., y0:y1, x0:x1] = masks
    return out


def is_box_near_crop_edge(boxes: torch.Tensor, crop_box: List[int], orig_box: List[int], atol: float = 20.0) -> torch.Tensor:
    """Boxes touching a crop edge that is not also an image edge (amg.py:78-88); [N,4] integer boxes -> bool [N]."""
    b = uncrop_boxes_xyxy(boxes, crop_box).float().cpu().numpy()
    near_crop = np.abs(b - np.asarray(crop_box, np.float32)[None, :]) <= atol
    near_image = np.abs(b - np.asarray(orig_box, np.float32)[None, :]) <= atol
    return torch.from_numpy(np.logical_and(near_crop, ~near_image).any(axis=1)).to(boxes.device)


def rle_to_mask(rle: Dict[str, Any]) -> np.ndarray:  # `counts` may be a list (reference format) or an integer array
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], dtype=np.int64)
    flat = np.repeat(np.arange(len(counts)) % 2 == 1, counts)  # runs alternate 0,1,0,... starting with a 0-run
    return flat.reshape(w, h).transpose()


def area_from_rle(rle: Dict[str, Any]) -> int:
    return int(np.sum(np.asarray(rle["counts"][1::2], dtype=np.int64)))


def remove_small_regions(mask: np.ndarray, area_thresh: float, mode: str) -> Tuple[np.ndarray, bool]:
    """Remove small disconnected regions ("islands") or fill small holes ("holes") of a boolean mask; returns (mask, modified)
    (utils/amg.py:267-291).  The reference is a host (numpy) helper built on cv2.connectedComponentsWithStats(mask, 8), which is
    not installed here; the 8-connected labelling comes from scipy.ndimage.label instead.  Only region sizes and membership are
    used, and both labellers number regions in raster order of their first pixel, so the largest-region tie-break is the same."""
    from scipy import ndimage
    assert mode in ["holes", "islands"]
    correct_holes = mode == "holes"
    mask = np.asarray(mask).astype(bool)
    working_mask = correct_holes ^ mask
    regions, n_regions = ndimage.label(working_mask, structure=np.ones((3, 3), dtype=np.uint8))
    n_labels = n_regions + 1                                   # label 0 is the background, as in cv2's stats row 0
    sizes = np.bincount(regions.reshape(-1), minlength=n_labels)[1:]
    small_regions = [i + 1 for i, s_ in enumerate(sizes) if s_ < area_thresh]
    if len(small_regions) == 0:
        return mask, False
    fill_labels = [0] + small_regions
    if not correct_holes:
        fill_labels = [i for i in range(n_labels) if i not in fill_labels]
        if len(fill_labels) == 0:                              # every region is below the threshold: keep the largest
            fill_labels = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill_labels), True


def coco_encode_rle(uncompressed_rle: Dict[str, Any]) -> Dict[str, Any]:
    """Uncompressed column-major RLE {"size": [h, w], "counts": [...]} -> COCO's compressed form with a utf-8 `counts` string
    (utils/amg.py:294-300 calls pycocotools.mask.frPyObjects, absent here).  Restates the published encoder (cocoapi maskApi.c
    rleToString): from the third run on, a run is stored as the difference to the run two places back; each value is emitted in
    5-bit groups, low group first, bit 5 = continuation, bit 4 of the last group = sign, offset by 48 into printable ASCII."""
    h, w = uncompressed_rle["size"]
    cnts = [int(c) for c in uncompressed_rle["counts"]]
    out = []
    for i, x in enumerate(cnts):
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return {"size": [int(h), int(w)], "counts": "".join(out)}


def coco_decode_rle(rle: Dict[str, Any]) -> Dict[str, Any]:
    """Inverse of coco_encode_rle (cocoapi maskApi.c rleFrString): compressed string -> uncompressed counts."""
    s_ = rle["counts"]
    s_ = s_.decode("utf-8") if isinstance(s_, (bytes, bytearray)) else s_
    cnts: List[int] = []
    p = 0
    while p < len(s_):
        x, k, more = 0, 0, True
        while more:
            c = ord(s_[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return {"size": list(rle["size"]), "counts": cnts}


# ---- device work ----------------------------------------------------------------------------------------------------
def calculate_stability_score(masks: torch.Tensor, mask_threshold: float, threshold_offset: float) -> torch.Tensor:
    """IoU between the mask logits thresholded at +offset and -offset (amg.py:156-176); fp32 [..., H, W] -> fp32 [...]."""
    m = _chk(masks.float().contiguous(), "masks", torch.float32)
    lead = m.shape[:-2]
    per = m.shape[-1] * m.shape[-2]
    N = m.numel() // per if per else 0
    counts = torch.empty((max(N, 1), 2), dtype=torch.int32, device=m.device)
    score = torch.empty((N,), dtype=torch.float32, device=m.device)
    _lib.call("ullsam_stability_score", m.data_ptr(), N, per, float(mask_threshold), float(threshold_offset), counts.data_ptr(),
              score.data_ptr(), _stream())
    return score.reshape(lead)


def threshold_masks(masks: torch.Tensor, mask_threshold: float) -> torch.Tensor:
    """masks > mask_threshold as uint8 (the binarisation step between logits and the integer helpers)."""
    m = _chk(masks.float().contiguous(), "masks", torch.float32)
    out = torch.empty(m.shape, dtype=torch.uint8, device=m.device)
    _lib.call("ullsam_threshold_u8", m.data_ptr(), out.data_ptr(), m.numel(), float(mask_threshold), _stream())
    return out


def _as_u8(masks: torch.Tensor) -> torch.Tensor:
    if masks.dtype == torch.bool:
        masks = masks.view(torch.uint8) if masks.is_contiguous() else masks.contiguous().view(torch.uint8)
    return _chk(masks.contiguous(), "masks", torch.uint8)


def batched_mask_to_box(masks: torch.Tensor) -> torch.Tensor:
    """XYXY boxes around binary masks, [0,0,0,0] for empty ones (amg.py:303-346); [..., H, W] -> int64 [..., 4]."""
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4, device=masks.device)
    m = _as_u8(masks)
    H, W = m.shape[-2:]
    N = m.numel() // (H * W)
    out = torch.empty((N, 4), dtype=torch.int32, device=m.device)
    _lib.call("ullsam_mask_to_box", m.data_ptr(), N, H, W, out.data_ptr(), _stream())
    out = out.long()
    return out.reshape(*masks.shape[:-2], 4) if masks.dim() > 2 else out[0]


def _rle_emit_launch(words: torch.Tensor, select, counts: np.ndarray, h: int, w: int, stage: dict):
    """First half of `_rles_from_words` for the generator's pipelined loop: launch the emission of the ordered change positions of the masks `select` and
    their copy back, all asynchronous -- offsets / selection go up from, and the positions come back into, the reusable PINNED buffers of `stage` (one dict
    per batch in flight).  -> (positions as a pinned int32 view, offsets, event to wait on before reading them)."""
    dev = words.device
    c = np.asarray(counts, dtype=np.int64)
    b = len(c)
    offs = np.concatenate([[0], np.cumsum(c)])
    n = max(int(offs[-1]), 1)
    up64 = _pinned(stage, "up64", b, torch.int64)
    up32 = _pinned(stage, "up32", b, torch.int32)
    up64[:b] = torch.from_numpy(offs[:-1].copy())
    up32[:b] = torch.from_numpy(np.asarray(select, dtype=np.int32))
    offs_d = up64[:b].to(dev, non_blocking=True)
    sel_d = up32[:b].to(dev, non_blocking=True)
    pos = torch.empty((n,), dtype=torch.int32, device=dev)
    _lib.call("ullsam_rle_emit", words.data_ptr(), sel_d.data_ptr(), b, h, w, offs_d.data_ptr(), pos.data_ptr(), _stream())
    pos_h = _pinned(stage, "pos", n, torch.int32)[:n]
    pos_h.copy_(pos, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    stage["_keep"] = (offs_d, sel_d, pos)     # the device buffers live until the slot is reused (their kernels have long run by then)
    return pos_h, offs, ev


def _pinned(stage: dict, key: str, n: int, dtype) -> torch.Tensor:
    t = stage.get(key)
    if t is None or t.numel() < n or t.dtype != dtype:
        t = torch.empty((max(n, 1024),), dtype=dtype, pin_memory=True)
        stage[key] = t
    return t


def _rles_from_words(words: torch.Tensor, select, counts: np.ndarray, first: np.ndarray, h: int, w: int,
                     as_list: bool = True) -> List[Dict[str, Any]]:
    """Change words [M, ceil(h/64), w] -> RLE dicts of the masks `select` (None = all), given their change counts / first bits.
    `as_list=False` leaves `counts` as int64 arrays (the generator converts only the records that survive NMS)."""
    b = len(counts)
    if b == 0:
        return []
    dev = words.device
    c = np.asarray(counts, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(c)])
    pos = torch.empty((max(int(offs[-1]), 1),), dtype=torch.int32, device=dev)
    offs_d = torch.from_numpy(offs[:-1].copy()).to(dev)
    sel_d = None if select is None else torch.as_tensor(np.asarray(select, dtype=np.int32), device=dev)
    _lib.call("ullsam_rle_emit", words.data_ptr(), None if sel_d is None else sel_d.data_ptr(), b, h, w, offs_d.data_ptr(),
              pos.data_ptr(), _stream())
    return _rle_records(pos.cpu().numpy().astype(np.int64), offs, first, h, w, as_list)


def _rle_records(pos_h: np.ndarray, offs: np.ndarray, first: np.ndarray, h: int, w: int, as_list: bool = True) -> List[Dict[str, Any]]:
    """Second half of `_rles_from_words`: run lengths = differences of the ordered change positions (host)."""
    out = []
    for i in range(len(offs) - 1):
        edges = np.concatenate([[0, 0] if first[i] else [0], pos_h[offs[i]:offs[i + 1]] + 1, [h * w]])
        runs = edges[1:] - edges[:-1]
        out.append({"size": [h, w], "counts": runs.tolist() if as_list else runs})
    return out


def mask_to_rle_pytorch(tensor: torch.Tensor) -> List[Dict[str, Any]]:
    """Uncompressed column-major RLE per mask, pycocotools layout (amg.py:107-135).  One coalesced pass packs the mask into per-column
    change words, a second emits the ordered change positions; the run lengths are their differences."""
    m = _as_u8(tensor)
    b, h, w = m.shape
    if b == 0:
        return []
    words = torch.empty((b, (h + 63) // 64, w), dtype=torch.int64, device=m.device)
    counts = torch.empty((b,), dtype=torch.int32, device=m.device)
    first = torch.empty((b,), dtype=torch.uint8, device=m.device)
    _lib.call("ullsam_rle_pack", m.data_ptr(), b, h, w, words.data_ptr(), counts.data_ptr(), first.data_ptr(), _stream())
    return _rles_from_words(words, None, counts.cpu().numpy(), first.cpu().numpy(), h, w)


class PostprocessedMasks:
    """Result of `postprocess_low_res`: per low-res map the stability counts, the box (crop coordinates), and the RLE change words
    of the mask placed in the full frame; `rles(select)` finishes the RLE of the chosen maps."""

    def __init__(self, words, scalars, frame):
        """scalars: ONE int32 device buffer [8, M] = rle_counts | first (bytes in its first M) | boxes [M, 4] | stability counts [M, 2]: one device -> host copy per batch."""
        self.words, self.frame = words, frame
        M = words.shape[0]
        h = scalars if isinstance(scalars, np.ndarray) else scalars.cpu().numpy()   # (a host array: the generator's pipelined loop copied the buffer back asynchronously)
        self.rle_counts = h[0][:M].copy()
        self.first = h[1].view(np.uint8)[:M].copy()
        self.boxes = h[2:6].reshape(-1)[:4 * M].reshape(M, 4).astype(np.int64)
        st = h[6:8].reshape(-1)[:2 * M].reshape(M, 2)
        with np.errstate(divide="ignore", invalid="ignore"):  # int32 / int32 true division -> fp32, as calculate_stability_score
            self.stability_score = st[:, 0].astype(np.float32) / st[:, 1].astype(np.float32)

    def rles(self, select, as_list: bool = True) -> List[Dict[str, Any]]:
        select = np.asarray(select, dtype=np.int64)
        return _rles_from_words(self.words, select, self.rle_counts[select], self.first[select], self.frame[0], self.frame[1], as_list)


def postprocess_low_res(low: torch.Tensor, img_size: int, input_size, crop_box, orig_size, mask_threshold: float,
                        threshold_offset: float, defer: bool = False) -> PostprocessedMasks:
    """Generator fast path over low-res logits [M, h, w]: Sam.postprocess_masks (sam.py:154-162) + calculate_stability_score +
    batched_mask_to_box + mask_to_rle_pytorch(uncrop_masks(...)) fused in one kernel; the full-resolution logits and masks are never
    written.  Equivalent to calling those helpers one after the other (same arithmetic per pixel)."""
    low = _chk(low if (low.dtype == torch.float32 and low.is_contiguous()) else low.float().contiguous(), "low_res_masks", torch.float32)
    M, lh, lw = low.shape
    x0, y0, x1, y1 = (int(v) for v in crop_box)
    fh, fw = int(orig_size[0]), int(orig_size[1])
    dev = low.device
    words = torch.empty((M, (fh + 63) // 64, fw), dtype=torch.int64, device=dev)
    sc = torch.empty((8, -(-M // 4) * 4), dtype=torch.int32, device=dev)   # rle_counts | first | boxes | stability counts: one buffer (rows 16-byte aligned), one copy back
    rle_counts, first, boxes, stab = sc[0], sc[1], sc[2:6], sc[6:8]
    _lib.call("ullsam_amg_postprocess", low.data_ptr(), None, M, lh, lw, int(img_size), int(input_size[0]), int(input_size[1]),
              y1 - y0, x1 - x0, fh, fw, x0, y0, float(mask_threshold), float(threshold_offset), words.data_ptr(),
              rle_counts.data_ptr(), first.data_ptr(), boxes.data_ptr(), stab.data_ptr(), _stream())
    if defer:
        return words, sc, (fh, fw)          # nothing read back yet: PostprocessedMasks(words, <host copy of sc>, frame) finishes it
    return PostprocessedMasks(words, sc, (fh, fw))


def box_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.nms semantics (the reference's AMG depends on torchvision, which is absent): kept indices in decreasing
    score order.  Ordering of the few thousand scores and the greedy scan are host bookkeeping; the N x N IoU suppression
    matrix is computed on the GPU."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    sc = scores.detach().float().cpu().numpy()
    order = np.lexsort((np.arange(n), -sc.astype(np.float64)))
    sorted_boxes = _chk(boxes.float()[torch.from_numpy(order).to(boxes.device)].contiguous(), "boxes", torch.float32)
    nw = (n + 63) // 64
    mask = torch.empty((n, nw), dtype=torch.int64, device=boxes.device)
    _lib.call("ullsam_nms_mask", sorted_boxes.data_ptr(), n, float(iou_threshold), mask.data_ptr(), _stream())
    mh = mask.cpu().numpy().view(np.uint64)
    removed = np.zeros(nw, dtype=np.uint64)
    keep = []
    for i in range(n):
        if not (removed[i >> 6] >> np.uint64(i & 63)) & np.uint64(1):
            keep.append(order[i])
            removed |= mh[i]
    return torch.from_numpy(np.asarray(keep, dtype=np.int64)).to(boxes.device)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.batched_nms: boxes of different categories never suppress each other (coordinate-offset trick)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coord = boxes.max()
    offsets = idxs.to(boxes) * (max_coord + 1)
    return box_nms(boxes + offsets[:, None], scores, iou_threshold)
