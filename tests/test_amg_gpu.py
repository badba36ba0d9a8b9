"""Automatic-mask-generation row: helper kernels bit-exact vs vectors captured from the reference's
utils/amg.py functions; box NMS vs the oracle's torchvision-semantics restatement; the generator driver vs a numpy re-statement of
the same flow over the oracle."""
import numpy as np
import pytest
import torch

from oracle import amg_oracle as AO
from oracle import ullsam_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def test_amg_helpers_bit_exact_vs_reference_vectors():
    from ullsam_amd.utils import amg as A
    g = U.gold("amg")
    logits = T(g["logits"])
    s = A.calculate_stability_score(logits, 0.0, 1.0).cpu().numpy()
    assert np.array_equal(s, g["stability"], equal_nan=True)
    s = A.calculate_stability_score(logits.reshape(3, 4, 96, 128), 0.25, 0.5).cpu().numpy()
    assert s.shape == (3, 4) and np.array_equal(s, g["stability_b"], equal_nan=True)
    binm = A.threshold_masks(logits, 0.0)
    assert np.array_equal(binm.cpu().numpy().astype(bool), g["logits"] > 0)
    boxes = A.batched_mask_to_box(binm)
    assert boxes.dtype == torch.int64 and np.array_equal(boxes.cpu().numpy(), g["boxes"])
    assert np.array_equal(A.batched_mask_to_box(binm.bool().reshape(3, 4, 96, 128)).cpu().numpy(), g["boxes_4d"])
    rles = A.mask_to_rle_pytorch(binm)
    assert [len(r["counts"]) for r in rles] == g["rle_lens"].tolist()
    assert np.concatenate([np.asarray(r["counts"], np.int64) for r in rles]).tolist() == g["rle_counts"].tolist()
    assert [A.area_from_rle(r) for r in rles] == g["rle_area"].tolist()
    assert all(r["size"] == [96, 128] for r in rles)
    for i, r in enumerate(rles):  # encode -> decode round trip
        assert np.array_equal(A.rle_to_mask(r), g["logits"][i] > 0)
    crop, orig = [100, 50, 228, 146], [0, 0, 400, 300]
    assert np.array_equal(A.is_box_near_crop_edge(boxes, crop, orig).cpu().numpy(), g["near_edge"])
    assert np.array_equal(A.is_box_near_crop_edge(boxes, [0, 0, 128, 96], [0, 0, 128, 96]).cpu().numpy(), g["near_edge_full"])
    assert np.array_equal(A.uncrop_boxes_xyxy(boxes, crop).cpu().numpy(), g["uncrop_boxes"])
    assert np.array_equal(A.uncrop_masks(binm, crop, 300, 400).sum((-1, -2)).cpu().numpy(), g["uncrop_masks_sum"])
    assert A.box_xyxy_to_xywh(torch.tensor([10, 20, 50, 80])).tolist() == g["xywh"].tolist()


def test_rle_full_size_and_edge_cases():
    from ullsam_amd.utils import amg as A
    rng = np.random.default_rng(0)
    m = rng.random((5, 1024, 1024)) > 0.5          # worst case: ~half a million runs per mask
    m[1] = False
    m[2] = True
    m[3, :, :512] = True; m[3, :, 512:] = False
    rles = A.mask_to_rle_pytorch(T(m))
    ref = AO.mask_to_rle(m)
    for a, b in zip(rles, ref):
        assert a["size"] == b["size"] and a["counts"] == b["counts"]
    assert rles[1]["counts"] == [1024 * 1024] and rles[2]["counts"] == [0, 1024 * 1024]
    assert A.mask_to_rle_pytorch(T(m[:0])) == []
    assert np.array_equal(A.batched_mask_to_box(T(m)).cpu().numpy(), AO.batched_mask_to_box(m))


@pytest.mark.parametrize("shape", [(3, 96, 128), (2, 70, 48), (2, 65, 33), (3, 1, 17), (3, 50, 1), (2, 64, 16), (2, 129, 1040), (1, 1, 1)])
def test_rle_and_box_ragged_shapes(shape):
    """Vector path (W % 16 == 0), scalar path, partial 64-row blocks, single rows / columns; bytes other than 0/1 count as set."""
    from ullsam_amd.utils import amg as A
    rng = np.random.default_rng(sum(shape))
    for density in (0.5, 0.03):
        m = rng.random(shape) < density
        m[0, -1, -1] = True                          # the last element of the flattening has no successor
        u8 = (m * rng.integers(1, 256, shape)).astype(np.uint8)
        for dev_in in (T(m), T(u8)):
            rles = A.mask_to_rle_pytorch(dev_in)
            for a, b in zip(rles, AO.mask_to_rle(m)):
                assert a["size"] == b["size"] and a["counts"] == b["counts"]
            assert np.array_equal(A.batched_mask_to_box(dev_in).cpu().numpy(), AO.batched_mask_to_box(m))
        for i, a in enumerate(rles):
            assert np.array_equal(A.rle_to_mask(a), m[i])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 700])
def test_box_nms_matches_oracle(n):
    from ullsam_amd.utils import amg as A
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 900, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 300, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = rng.random(n).astype(np.float32)
    scores[: n // 3] = scores[0]  # ties: lower index first
    keep = A.box_nms(T(boxes), T(scores), 0.7).cpu().numpy()
    ref = AO.box_nms(boxes, scores, 0.7)
    assert keep.tolist() == ref.tolist()
    kb = boxes[keep]  # property: survivors do not suppress each other
    for i in range(len(kb)):
        iw = np.maximum(np.minimum(kb[i, 2], kb[:, 2]) - np.maximum(kb[i, 0], kb[:, 0]), 0)
        ih = np.maximum(np.minimum(kb[i, 3], kb[:, 3]) - np.maximum(kb[i, 1], kb[:, 1]), 0)
        iou = iw * ih / (AO.box_area(kb)[i] + AO.box_area(kb) - iw * ih)
        iou[i] = 0
        assert (iou <= 0.7 + 1e-6).all()
    cat = T((np.arange(n) % 2).astype(np.int64))
    kc = A.batched_nms(T(boxes), T(scores), cat, 0.7).cpu().numpy()
    assert set(kc.tolist()) >= set(keep.tolist())  # per-category NMS can only keep more


def _small_sam():
    from ullsam_amd.build_sam import _build_sam
    from tests.test_model_gpu import load
    P = {}
    P.update(U.vit_params(U.VIT_SMALL, 0, "image_encoder."))
    P.update(O.fill_state(O.prompt_encoder_shapes(prefix="prompt_encoder."), 0))
    P.update(O.fill_state(O.mask_decoder_shapes(prefix="mask_decoder."), 0))
    return load(_build_sam(128, 2, 2, [1]), P), P


def _smooth_logits(rng, m, h, w, cells=6):
    """Low-frequency random fields: masks with blobs and long borders instead of salt-and-pepper."""
    coarse = rng.standard_normal((m, cells, cells)).astype(np.float32) * 3
    return O.bilinear_resize(coarse, (h, w)) + rng.standard_normal((m, h, w)).astype(np.float32) * 0.05


@pytest.mark.parametrize("case", [
    dict(low=(5, 256, 256), S=1024, inp=(1024, 1024), crop=[0, 0, 2048, 2048], orig=(2048, 2048)),     # BASELINE configs[4] shape
    dict(low=(4, 256, 256), S=1024, inp=(768, 1024), crop=[0, 0, 512, 384], orig=(384, 512)),           # downscale, padded input
    dict(low=(4, 64, 64), S=256, inp=(200, 256), crop=[30, 20, 30 + 333, 20 + 260], orig=(300, 400)),   # crop inside a larger frame
    dict(low=(3, 64, 64), S=256, inp=(256, 190), crop=[0, 7, 95, 7 + 128], orig=(135, 95)),             # crop flush with three frame edges
    dict(low=(3, 32, 32), S=128, inp=(128, 128), crop=[0, 0, 70, 70], orig=(70, 70)),                   # partial 64-row block
])
def test_fused_postprocess_equals_the_helper_chain(case):
    """postprocess_low_res == postprocess_masks -> calculate_stability_score -> threshold -> batched_mask_to_box -> uncrop_masks ->
    mask_to_rle_pytorch, the chain the reference's helpers were written for."""
    from ullsam_amd import ops
    from ullsam_amd.utils import amg as A
    rng = np.random.default_rng(7)
    m, lh, lw = case["low"]
    low = T(_smooth_logits(rng, m, lh, lw))
    low[0] = -5.0                                     # empty mask
    low[1] = 5.0                                      # full mask
    S, inp, crop, orig = case["S"], case["inp"], case["crop"], case["orig"]
    ch, cw = crop[3] - crop[1], crop[2] - crop[0]
    thr, off = 0.0, 0.5
    up, _ = ops.resize_bilinear(low, (S, S))
    logits, _ = ops.resize_bilinear(up, (ch, cw), valid_hw=inp)
    stab = A.calculate_stability_score(logits, thr, off).cpu().numpy()
    binm = A.threshold_masks(logits, thr)
    boxes = A.batched_mask_to_box(binm).cpu().numpy()
    rles = A.mask_to_rle_pytorch(A.uncrop_masks(binm, crop, orig[0], orig[1]))
    pp = A.postprocess_low_res(low, S, inp, crop, orig, thr, off)
    assert np.array_equal(pp.boxes, boxes)
    assert np.array_equal(pp.stability_score, stab, equal_nan=True)
    got = pp.rles(np.arange(m))
    for a, b in zip(got, rles):
        assert a == b
    sub = pp.rles([m - 1, 1])
    assert sub[0] == rles[m - 1] and sub[1] == rles[1]
    # and against the numpy oracle (independent arithmetic: border pixels may flip)
    ref = O.bilinear_resize(O.bilinear_resize(low.cpu().numpy(), (S, S))[..., :inp[0], :inp[1]], (ch, cw)) > thr
    for i in range(m):
        full = np.zeros(orig, bool)
        full[crop[1]:crop[3], crop[0]:crop[2]] = ref[i]
        assert (AO.rle_to_mask(got[i]) != full).mean() < 1e-4


@pytest.mark.parametrize("fused", [True, False])
def test_generator_matches_numpy_flow_over_the_oracle(fused):
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    sam, P = _small_sam()
    img = U.rand_image((3, 384, 512), 21, 255.0)
    iou_thr, stab_thr, stab_off, nms_thr, side = -1e3, 0.5, 0.05, 1.0, 6  # random-weight masks are full-image blobs: NMS itself is pinned in test_box_nms_matches_oracle
    gen = SamAutomaticMaskGenerator(sam, points_per_side=side, points_per_batch=20, pred_iou_thresh=iou_thr,
                                    stability_score_thresh=stab_thr, stability_score_offset=stab_off, box_nms_thresh=nms_thr,
                                    output_mode="uncompressed_rle", fused_postprocess=fused)
    got = gen.generate(torch.from_numpy(img))
    # ---- the same flow in numpy over the oracle
    h, w = 384, 512
    nh, nw = 768, 1024
    x = O.bilinear_resize(img, (nh, nw))
    mean = np.asarray([123.675, 116.28, 103.53], np.float32).reshape(3, 1, 1)
    std = np.asarray([58.395, 57.12, 57.375], np.float32).reshape(3, 1, 1)
    x = np.pad((x - mean) / std, ((0, 0), (0, 1024 - nh), (0, 0)))
    emb = O.vit_encoder(x[None], P, prefix="image_encoder.", **U.vit_run_cfg(U.VIT_SMALL))
    pts = AO.build_point_grid(side) * np.array([[w, h]])
    coords = (pts * np.array([[nw / w, nh / h]])).astype(np.float32)[:, None, :]
    sp, de = O.prompt_encoder(P, (coords, np.ones((len(pts), 1), np.int64)), None, None, prefix="prompt_encoder.")
    low, iou = O.mask_decoder(P, emb, O.dense_pe(P, prefix="prompt_encoder."), sp, de, True, prefix="mask_decoder.")
    low, iou = low.reshape(-1, 256, 256), iou.reshape(-1)
    k = iou > iou_thr
    low, iou_k, pk = low[k], iou[k], np.repeat(pts, 3, 0)[k]
    masks = O.bilinear_resize(O.bilinear_resize(low, (1024, 1024))[..., :nh, :nw], (h, w))
    stab = AO.calculate_stability_score(masks, 0.0, stab_off)
    k2 = stab >= stab_thr
    masks, iou_k, stab, pk = masks[k2], iou_k[k2], stab[k2], pk[k2]
    boxes = AO.batched_mask_to_box(masks > 0)
    keep = AO.box_nms(boxes.astype(np.float32), iou_k, nms_thr)
    assert len(got) == len(keep) and len(keep) >= 2, (len(got), len(keep))
    for rec, j in zip(got, keep):
        assert abs(rec["predicted_iou"] - float(iou_k[j])) < 1e-4 and abs(rec["stability_score"] - float(stab[j])) < 2e-3
        assert rec["point_coords"][0] == pytest.approx(pk[j].tolist())
        m_ref = masks[j] > 0
        m_got = AO.rle_to_mask(rec["segmentation"])
        assert O.calc_iou(m_got, m_ref) > 0.999, O.calc_iou(m_got, m_ref)
        assert rec["area"] == int(m_got.sum())
        bx = AO.batched_mask_to_box(m_got[None])[0]
        assert rec["bbox"] == [int(bx[0]), int(bx[1]), int(bx[2] - bx[0]), int(bx[3] - bx[1])]
        assert rec["crop_box"] == [0, 0, w, h]


def test_generator_with_crops_invariants():
    """Two crop layers: records stay consistent (area / bbox / RLE), cross-crop NMS leaves no pair above the threshold."""
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    sam, _ = _small_sam()
    img = U.rand_image((3, 600, 800), 22, 255.0)
    gen = SamAutomaticMaskGenerator(sam, points_per_side=4, points_per_batch=64, pred_iou_thresh=-1e3, stability_score_thresh=0.5,
                                    stability_score_offset=0.05, crop_n_layers=1, crop_n_points_downscale_factor=2)
    recs = gen.generate(torch.from_numpy(img))
    assert len(recs) > 0
    boxes = []
    for r in recs:
        seg = r["segmentation"]
        assert seg.shape == (600, 800) and seg.dtype == bool and r["area"] == int(seg.sum())
        b = AO.batched_mask_to_box(seg[None])[0]
        assert r["bbox"] == [int(b[0]), int(b[1]), int(b[2] - b[0]), int(b[3] - b[1])]
        assert r["stability_score"] >= 0.5
        boxes.append([b[0], b[1], b[2], b[3]])
    kb = np.asarray(boxes, np.float32)
    for i in range(len(kb)):
        iw = np.maximum(np.minimum(kb[i, 2], kb[:, 2]) - np.maximum(kb[i, 0], kb[:, 0]), 0)
        ih = np.maximum(np.minimum(kb[i, 3], kb[:, 3]) - np.maximum(kb[i, 1], kb[:, 1]), 0)
        iou = iw * ih / np.maximum(AO.box_area(kb)[i] + AO.box_area(kb) - iw * ih, 1e-9)
        iou[i] = 0
        assert (iou <= 0.7 + 1e-6).all()


def test_generator_fused_and_helper_chain_agree_with_crops():
    """crop_n_layers=1 (five crops, masks uncropped into the frame): the fused post-processing path and the helper chain return the
    same records."""
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    sam, _ = _small_sam()
    img = torch.from_numpy(U.rand_image((3, 600, 800), 22, 255.0))
    kw = dict(points_per_side=4, points_per_batch=64, pred_iou_thresh=-1e3, stability_score_thresh=0.5, stability_score_offset=0.05,
              crop_n_layers=1, crop_n_points_downscale_factor=2, output_mode="uncompressed_rle")
    a = SamAutomaticMaskGenerator(sam, fused_postprocess=True, **kw).generate(img)
    b = SamAutomaticMaskGenerator(sam, fused_postprocess=False, **kw).generate(img)
    assert len(a) == len(b) > 0
    for ra, rb in zip(a, b):
        assert ra["segmentation"] == rb["segmentation"] and ra["bbox"] == rb["bbox"] and ra["area"] == rb["area"]
        assert ra["crop_box"] == rb["crop_box"] and ra["point_coords"] == rb["point_coords"]
        assert ra["stability_score"] == rb["stability_score"] and ra["predicted_iou"] == rb["predicted_iou"]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generator_reuses_the_image_side_across_point_batches(dtype):
    """The decoder's prompt-independent image side (keys = embedding + dense prompt, their model-dtype copies, layer 0's K / V projections --
    transformer.py:220-242 recomputes them for every call) is computed by the first point batch of a crop and reused by the others
    (`reuse_image_side`, default): same records as recomputing it per batch, over several batches per crop and two crop layers."""
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    sam, _ = _small_sam()
    sam = sam.to(dtype)
    img = torch.from_numpy(U.rand_image((3, 600, 800), 23, 255.0))
    kw = dict(points_per_side=6, points_per_batch=8, pred_iou_thresh=-1e3, stability_score_thresh=0.5, stability_score_offset=0.05,
              crop_n_layers=1, crop_n_points_downscale_factor=2, output_mode="uncompressed_rle")
    ga, gb = SamAutomaticMaskGenerator(sam, **kw), SamAutomaticMaskGenerator(sam, **kw)
    gb.reuse_image_side = False
    a, b = ga.generate(img), gb.generate(img)
    assert len(a) == len(b) > 0
    for ra, rb in zip(a, b):
        assert ra["segmentation"] == rb["segmentation"] and ra["bbox"] == rb["bbox"] and ra["area"] == rb["area"]
        assert ra["stability_score"] == rb["stability_score"] and ra["predicted_iou"] == rb["predicted_iou"] and ra["point_coords"] == rb["point_coords"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generator_records_do_not_depend_on_points_per_batch(dtype):
    """SAM's `points_per_batch` is a memory knob (0.123 / 0.103 / 0.095 s per real-size tile at 64 / 128 / 256): the records of a crop must be the same
    whichever way its point grid is cut into decoder batches -- every decoder kernel works per prompt (token kernels: a workgroup per prompt; image-side kernels:
    rows of one prompt), and the launches whose tile shape depends on the row count keep their k order.  Masks, boxes, areas, stability scores: bit-equal; the
    predicted IoU to its last place (see below)."""
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    sam, _ = _small_sam()
    sam = sam.to(dtype)
    img = torch.from_numpy(U.rand_image((3, 600, 800), 29, 255.0))
    kw = dict(points_per_side=6, pred_iou_thresh=-1e3, stability_score_thresh=0.5, stability_score_offset=0.05, output_mode="uncompressed_rle")
    recs = [SamAutomaticMaskGenerator(sam, points_per_batch=ppb, **kw).generate(img) for ppb in (5, 12, 36)]
    assert len(recs[0]) > 0
    for other in recs[1:]:
        assert len(other) == len(recs[0])
        for ra, rb in zip(recs[0], other):
            assert ra["segmentation"] == rb["segmentation"] and ra["bbox"] == rb["bbox"] and ra["area"] == rb["area"]
            assert ra["stability_score"] == rb["stability_score"] and ra["point_coords"] == rb["point_coords"]
            # the IoU head's fp32 token-side linears pick their kernel by row count (one wave per output for a handful of rows, the row-block kernels beyond):
            # the predicted IoU may move in its last place (measured 6e-8), the masks and everything derived from them may not
            assert abs(ra["predicted_iou"] - rb["predicted_iou"]) <= 1e-6 * max(1.0, abs(ra["predicted_iou"]))


def test_generator_min_mask_region_area_and_coco_rle():
    """min_mask_region_area: every surviving mask has no island and no hole smaller than the threshold (remove_small_regions is
    idempotent on it), areas / boxes are recomputed for changed masks; output_mode="coco_rle" round-trips to the same masks."""
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    from ullsam_amd.utils import amg as A
    sam, _ = _small_sam()
    img = torch.from_numpy(U.rand_image((3, 512, 640), 23, 255.0))
    kw = dict(points_per_side=4, points_per_batch=64, pred_iou_thresh=-1e3, stability_score_thresh=0.3, stability_score_offset=0.05)
    plain = SamAutomaticMaskGenerator(sam, **kw).generate(img)
    clean = SamAutomaticMaskGenerator(sam, min_mask_region_area=200, **kw).generate(img)
    assert 0 < len(clean) <= len(plain)
    for r in clean:
        seg = r["segmentation"]
        for mode in ("holes", "islands"):
            again, changed = A.remove_small_regions(seg, 200, mode)
            assert not changed or np.array_equal(again, seg)   # (only the "keep the largest island" rule may report a change)
        assert r["area"] == int(seg.sum())
        b = AO.batched_mask_to_box(seg[None])[0]
        assert r["bbox"] == [int(b[0]), int(b[1]), int(b[2] - b[0]), int(b[3] - b[1])]
    coco = SamAutomaticMaskGenerator(sam, output_mode="coco_rle", **kw).generate(img)
    assert len(coco) == len(plain)
    for rc, rp in zip(coco, plain):
        assert isinstance(rc["segmentation"]["counts"], str)
        assert np.array_equal(A.rle_to_mask(A.coco_decode_rle(rc["segmentation"])), rp["segmentation"])


# ---- BASELINE configs[4] at its real size: 64x64 points on a 2048^2 tile, SAM ViT-H encoder, bf16 and fp8 ViT linears --------------------
# No weights exist offline, and a random-init decoder draws full-frame textures: every box is (nearly) the whole tile, so SAM's box NMS at 0.7
# would keep ONE record.  The real-size generator test therefore runs on the STRUCTURED decoder of ullsam_amd.utils.synthetic.blob_decoder_init
# (written-down weights under which a positive click draws a disc around itself, by SAM's own image -> token attention mechanism): 4096 clicks x 3
# discs of different radii with different boxes, predicted IoU 0.93 / 0.91 / 0.89, stability (offset 1.0) ~0.94 / 0.89 / 0.76 -- so the
# predicted-IoU filter (0.90), the stability filter (0.92) and box NMS at SAM's 0.7 all have work to do, and a few hundred records survive.
REAL_AMG = dict(points_per_side=64, points_per_batch=64, pred_iou_thresh=0.90, stability_score_thresh=0.92, stability_score_offset=1.0,
                box_nms_thresh=0.7, output_mode="uncompressed_rle")


def _box_iou_matrix(kb):
    x0 = np.maximum(kb[:, None, 0], kb[None, :, 0]); y0 = np.maximum(kb[:, None, 1], kb[None, :, 1])
    x1 = np.minimum(kb[:, None, 2], kb[None, :, 2]); y1 = np.minimum(kb[:, None, 3], kb[None, :, 3])
    inter = np.maximum(x1 - x0, 0) * np.maximum(y1 - y0, 0)
    area = (kb[:, 2] - kb[:, 0]) * (kb[:, 3] - kb[:, 1])
    return inter / np.maximum(area[:, None] + area[None, :] - inter, 1e-9)


def test_blob_decoder_draws_a_disc_around_the_click():
    """The structured decoder parameters (synthetic.blob_decoder_state) through the HIP prompt encoder + mask decoder: every click's three masks
    are discs centred on the click (centroid within 2 low-res pixels), nested in size, with the constant IoU predictions -- and the numpy
    oracle (the reference's algorithm) draws the same masks from the same parameters."""
    from ullsam_amd.utils.synthetic import blob_decoder_init
    sam, P = _small_sam()
    blob_decoder_init(sam)
    Pn = {k: v.detach().float().cpu().numpy() for k, v in sam.state_dict().items()}
    pts = np.array([[[300.0, 400.0]], [[800.0, 200.0]], [[512.0, 760.0]], [[620.0, 480.0]]], np.float32)   # (interior clicks: a disc clipped by the frame has its centroid off the click)
    lbl = np.ones((4, 1), np.int32)
    e = torch.zeros((1, 256, 64, 64), device=DEV)
    sp, de = sam.prompt_encoder(points=(torch.from_numpy(pts).to(DEV), torch.from_numpy(lbl).to(DEV)), boxes=None, masks=None)
    low, iou = sam.mask_decoder(image_embeddings=e, image_pe=sam.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sp,
                                dense_prompt_embeddings=de, multimask_output=True)
    low, iou = low.float().cpu().numpy(), iou.float().cpu().numpy()
    spo, deo = O.prompt_encoder(Pn, (pts, lbl), None, None, None, prefix="prompt_encoder.")
    lowo, iouo = O.mask_decoder(Pn, np.zeros((1, 256, 64, 64), np.float32), O.dense_pe(Pn, prefix="prompt_encoder."), spo, deo, True, prefix="mask_decoder.")
    assert np.abs(iou - np.asarray([0.93, 0.91, 0.89], np.float32)).max() < 1e-3 and np.abs(iou - iouo).max() < 1e-4
    for i in range(4):
        areas = []
        for k in range(3):
            m = low[i, k] > 0
            assert O.calc_iou(m, lowo[i, k] > 0) > 0.98, (i, k)
            ys, xs = np.nonzero(m)
            assert abs(xs.mean() - pts[i, 0, 0] / 4) < 2.5 and abs(ys.mean() - pts[i, 0, 1] / 4) < 2.5, (i, k, xs.mean(), ys.mean())
            assert (xs.max() - xs.min()) < 140 and (ys.max() - ys.min()) < 140          # a disc, not a full-frame texture
            areas.append(int(m.sum()))
        assert areas[0] > areas[1] > areas[2] > 500, areas


def test_generator_real_size_vit_h_2048_tile_with_box_nms():
    """`SamAutomaticMaskGenerator` exactly as configs[4] names it -- ViT-H encoder, one 2048^2 microscopy tile, 64 x 64 = 4096 point prompts
    in batches of 64 with multimask output (12288 candidate masks) -- on the structured (disc-drawing) decoder, with SAM's box NMS at 0.7:
      * the fused post-processing kernel and the helper chain (postprocess_masks -> calculate_stability_score -> threshold ->
        batched_mask_to_box -> mask_to_rle_pytorch, utils/amg.py:107-176,303-346) return the SAME records at full size;
      * >= 100 records survive; the predicted-IoU filter (the third mask of every click), the stability filter and NMS each removed candidates;
        no two kept boxes overlap by more than the NMS threshold; every record is consistent: area = RLE foreground, bbox = box of the decoded
        mask, scores above their thresholds, the click inside its own box."""
    import bench
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    from ullsam_amd.utils.synthetic import blob_decoder_init, microscopy_tile
    sam = blob_decoder_init(bench.build_model("h", "none", torch.bfloat16, DEV))
    img_np, _ = microscopy_tile(7, size=2048, n_cells=40, r_range=(90.0, 260.0))
    img = torch.from_numpy(img_np * 255.0).to(DEV)
    fused = SamAutomaticMaskGenerator(sam, fused_postprocess=True, **REAL_AMG).generate(img)
    chain = SamAutomaticMaskGenerator(sam, fused_postprocess=False, **REAL_AMG).generate(img)
    no_nms = SamAutomaticMaskGenerator(sam, fused_postprocess=True, **dict(REAL_AMG, box_nms_thresh=1.0)).generate(img)
    print(f"configs[4] real size: {len(fused)} records kept of 12288 candidates ({len(no_nms)} before box NMS at 0.7)")
    assert len(fused) == len(chain) and len(fused) >= 100, (len(fused), len(chain))
    assert len(fused) < len(no_nms) <= 4096              # NMS suppressed candidates; the IoU filter removed every click's third mask, the stability filter the second
    for ra, rb in zip(fused, chain):
        assert ra["segmentation"] == rb["segmentation"] and ra["bbox"] == rb["bbox"] and ra["area"] == rb["area"]
        assert ra["point_coords"] == rb["point_coords"] and ra["stability_score"] == rb["stability_score"] and ra["predicted_iou"] == rb["predicted_iou"]
    boxes = []
    for i, r in enumerate(fused):
        assert r["predicted_iou"] > REAL_AMG["pred_iou_thresh"] and r["stability_score"] >= REAL_AMG["stability_score_thresh"]
        assert r["segmentation"]["size"] == [2048, 2048] and sum(r["segmentation"]["counts"]) == 2048 * 2048
        assert r["area"] == sum(r["segmentation"]["counts"][1::2]) and r["crop_box"] == [0, 0, 2048, 2048]
        x, y, w, h = r["bbox"]
        boxes.append([x, y, x + w, y + h])
        px, py = r["point_coords"][0]
        assert x - 1 <= px <= x + w + 1 and y - 1 <= py <= y + h + 1 and w < 1400 and h < 1400, r["bbox"]
        if i % max(1, len(fused) // 16) == 0:       # decode a sample of the masks (4 MiB each) and re-derive the box
            seg = AO.rle_to_mask(r["segmentation"])
            b = AO.batched_mask_to_box(seg[None])[0]
            assert [int(b[0]), int(b[1]), int(b[2] - b[0]), int(b[3] - b[1])] == r["bbox"] and int(seg.sum()) == r["area"]
    iou = _box_iou_matrix(np.asarray(boxes, np.float32))
    np.fill_diagonal(iou, 0.0)
    assert float(iou.max()) <= REAL_AMG["box_nms_thresh"] + 1e-6, float(iou.max())


def test_generator_real_size_fp8_encoder_against_bf16():
    """configs[4]'s "fp8 MFMA ViT path" at real size: the generator with `fp8_linears=True` against the bf16 encoder on a RANDOM-init decoder (whose
    masks depend on the encoder's features through every attention; the structured decoder above barely looks at them).  A random decoder's boxes
    are all the whole tile, so NMS is off (1.0) here; the score thresholds are the 85th percentiles of the scores on a 16 x 16 probe grid, so that
    on the order of a hundred masks survive whatever the init draws.  The kept sets overlap (>= 80 % of the bf16 records have an fp8 record from
    the same click with box IoU >= 0.9) and matched masks agree (mask IoU >= 0.94, mean >= 0.96 on a sample: fp8 operands cost ~0.03 of mask IoU)."""
    import bench
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    from ullsam_amd.utils.synthetic import microscopy_tile
    sam = bench.build_model("h", "none", torch.bfloat16, DEV)
    img_np, _ = microscopy_tile(7, size=2048, n_cells=40, r_range=(90.0, 260.0))
    img = torch.from_numpy(img_np * 255.0).to(DEV)
    probe = SamAutomaticMaskGenerator(sam, points_per_side=16, points_per_batch=64, pred_iou_thresh=-1e3, stability_score_thresh=-1.0,
                                      stability_score_offset=0.1, box_nms_thresh=1.0, output_mode="uncompressed_rle").generate(img)
    kw = dict(points_per_side=64, points_per_batch=64, stability_score_offset=0.1, box_nms_thresh=1.0, output_mode="uncompressed_rle",
              pred_iou_thresh=float(np.percentile([r["predicted_iou"] for r in probe], 85)),
              stability_score_thresh=float(np.percentile([r["stability_score"] for r in probe], 85)))
    fused = SamAutomaticMaskGenerator(sam, fused_postprocess=True, **kw).generate(img)
    sam.image_encoder.fp8_linears = True
    f8 = SamAutomaticMaskGenerator(sam, fused_postprocess=True, **kw).generate(img)
    sam.image_encoder.fp8_linears = False
    assert len(fused) >= 20 and len(f8) >= 20, (len(fused), len(f8), kw)
    by_pt = {}
    for r in f8:
        by_pt.setdefault(tuple(r["point_coords"][0]), []).append(r)
    matched, pairs = 0, []
    for r in fused:
        x, y, w, h = r["bbox"]
        a = np.asarray([[x, y, x + w, y + h]], np.float32)
        best, best_r = 0.0, None
        for c in by_pt.get(tuple(r["point_coords"][0]), []):
            cx, cy, cw, ch = c["bbox"]
            v = float(_box_iou_matrix(np.concatenate([a, np.asarray([[cx, cy, cx + cw, cy + ch]], np.float32)]))[0, 1])
            if v > best:
                best, best_r = v, c
        if best >= 0.9:
            matched += 1
            pairs.append((r, best_r))
    share = matched / len(fused)
    ious = [O.calc_iou(AO.rle_to_mask(a["segmentation"]), AO.rle_to_mask(b["segmentation"])) for a, b in pairs[::max(1, len(pairs) // 12)]]
    print(f"fp8 vs bf16: {len(f8)} vs {len(fused)} records, {share:.3f} of the bf16 records matched by an fp8 record of the same click; "
          f"mask IoU of matched pairs min {min(ious):.4f} mean {float(np.mean(ious)):.4f}")
    assert share >= 0.8, share
    assert min(ious) >= 0.94 and float(np.mean(ious)) >= 0.96, ious     # the fp8 path's stated accuracy (README / DESIGN 7c): mask IoU ~0.97 vs bf16, a correctness demonstration


def test_generate_batch_equals_generate_per_tile_under_rccl_world_1():
    """generate_batch (tiles sharded over ranks, ragged records gathered) with an RCCL process group of one rank -- what one box offers -- returns
    exactly the per-tile `generate` results in tile order; the world-2 split is covered under gloo (tests/test_host_cpu.py)."""
    import os
    import torch.distributed as dist
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    from ullsam_amd.utils.synthetic import blob_decoder_init
    sam, _ = _small_sam()
    blob_decoder_init(sam)
    imgs = [torch.from_numpy(U.rand_image((3, 512, 640), 30 + i, 255.0)) for i in range(3)]
    gen = SamAutomaticMaskGenerator(sam, points_per_side=8, points_per_batch=64, pred_iou_thresh=0.90, stability_score_thresh=0.92, output_mode="uncompressed_rle")
    want = [gen.generate(im) for im in imgs]
    assert sum(len(w) for w in want) > 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29583")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        got = gen.generate_batch(imgs)
    finally:
        if own:
            dist.destroy_process_group()
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("P,T,shared", [(16, 7, True), (8, 9, False), (5, 16, True), (1, 7, False)])
def test_fused_image_to_token_block_equals_the_separate_launches(P, T, shared):
    """transformer.FUSED_I2T (default for bf16 from 1024 image tokens): the image -> token half of a two-way block -- q projection, attention over the
    T tokens, output projection + fp32 residual, norm4 with its three outputs -- as ONE kernel (csrc/decoder.hip i2t_block_kernel) against the five
    launches it replaces, through TwoWayTransformer.forward_tokens on random weights: same bf16 roundings at the same places, sums in another
    order.  Both image-side layouts: one image shared by all prompts (layer 0 broadcast) and one stream per prompt; a ragged prompt count."""
    from ullsam_amd.modeling import transformer as TR
    torch.manual_seed(P * 31 + T)
    tw = TR.TwoWayTransformer(depth=2, embedding_dim=256, num_heads=8, mlp_dim=2048).to(DEV)
    for prm in tw.parameters():
        torch.nn.init.normal_(prm, std=0.06)
    for n, prm in tw.named_parameters():
        if "norm" in n and n.endswith("weight"):
            torch.nn.init.normal_(prm, mean=1.0, std=0.1)
    tw = tw.to(torch.bfloat16)
    N = 4096
    keys = torch.randn((1 if shared else P, N, 256), device=DEV)
    key_pe = torch.randn((N, 256), device=DEV)
    tokens = torch.randn((P, T, 256), device=DEV)
    outs = []
    old = TR.FUSED_I2T
    try:
        for on in (True, False):
            TR.FUSED_I2T = on
            for kc in (False, True):
                q, k = tw.forward_tokens(keys, key_pe, tokens, keys_in_compute_dtype=kc)
                outs.append((q.float(), k.float()))
    finally:
        TR.FUSED_I2T = old
    torch.cuda.synchronize()
    for (qa, ka), (qb, kb) in zip(outs[:2], outs[2:]):
        assert torch.isfinite(qa).all() and torch.isfinite(ka).all()
        assert float((qa - qb).abs().max()) < 2e-2 * max(1.0, float(qb.abs().max())), float((qa - qb).abs().max())
        assert float((ka - kb).abs().max()) < 3e-2 * max(1.0, float(kb.abs().max())), float((ka - kb).abs().max())
        assert float((ka - kb).abs().mean()) < 2e-3 * max(1.0, float(kb.abs().mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [4096, 5 * 4096 + 7, 64 * 4096])
def test_fused_k_and_v_projection_of_the_image_side_equals_the_two_gemms(rows):
    """ops.kv_proj (csrc/decoder.hip kv_proj_kernel: both weight matrices resident in LDS, one pass over (keys + pe) and keys) against the two ops.gemm launches it
    replaces (the 128x128 tile kernel): fp32 sums of 256 products in another order, then the same rounding to bf16 -- a bf16 step of a value now and then, nothing more;
    and against float64 on the bf16-rounded operands.  A ragged row count exercises the last group's clamp."""
    from ullsam_amd import ops
    g = torch.Generator(device=DEV); g.manual_seed(rows)
    xk = torch.randn(rows, 256, device=DEV, generator=g).bfloat16()
    xv = torch.randn(rows, 256, device=DEV, generator=g).bfloat16()
    wk = (torch.randn(128, 256, device=DEV, generator=g) * 0.08).bfloat16()
    wv = (torch.randn(128, 256, device=DEV, generator=g) * 0.08).bfloat16()
    bk, bv = torch.randn(128, device=DEV, generator=g) * 0.1, torch.randn(128, device=DEV, generator=g) * 0.1
    K, V = ops.kv_proj(xk, xv, wk, bk, wv, bv)
    Kr, Vr = ops.gemm(xk, wk, bk), ops.gemm(xv, wv, bv)
    torch.cuda.synchronize()
    for got, ref, x, w, b in ((K, Kr, xk, wk, bk), (V, Vr, xv, wv, bv)):
        assert got.shape == ref.shape == (rows, 128) and torch.isfinite(got.float()).all()
        d = (got.float() - ref.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(ref.float().abs().max())) and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))
        want = x[:4096].double() @ w.double().T + b.double()
        assert float((got[:4096].double() - want).abs().max()) < 2.0 ** -8 * max(1.0, float(want.abs().max())) + 1e-6
    K2, V2 = ops.kv_proj(xk, xv, wk, None, wv, None)
    assert float((K2.float() - ops.gemm(xk, wk).float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(Kr.float().abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("P,nm", [(3, 4), (1, 4), (5, 1)])
def test_fused_second_upscaling_and_hypernetwork_product_equals_the_separate_launches(P, nm):
    """ops.up2_hyper_masks (second transposed convolution as Linear 64 -> 4 x 32, GELU, rounding to bf16, hypernetwork product: one kernel, the upscaled
    embedding never written) against ops.gemm(act=GELU) + ops.hyper_masks on the same operands: same roundings in the same places, the k sums of the
    product in another order (a bf16 step of a channel value now and then)."""
    from ullsam_amd import ops
    g = torch.Generator(device=DEV); g.manual_seed(P * 10 + nm)
    H = W = 64
    u1 = torch.randn(P * H * W * 4, 64, device=DEV, generator=g).bfloat16()
    w1 = (torch.randn(128, 64, device=DEV, generator=g) * 0.15).bfloat16()
    b1 = torch.randn(128, device=DEV, generator=g) * 0.1
    hyper = torch.randn(P, nm, 32, device=DEV, generator=g)
    ref = ops.hyper_masks(ops.gemm(u1, w1, b1, act=ops.ACT_GELU), hyper, P, nm, H, W, 32)
    got = ops.up2_hyper_masks(u1, w1, b1, hyper, P, nm, H, W)
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (P, nm, 256, 256) and torch.isfinite(got).all()
    d = (got - ref).abs()
    assert float(d.max()) < 2e-2 * max(1.0, float(ref.abs().max())) and float(d.mean()) < 2e-4 * max(1.0, float(ref.abs().mean())), (float(d.max()), float(d.mean()))
    # against fp64 on the bf16-rounded operands
    x = (u1.double() @ w1.double().T + b1.double())
    gel = (0.5 * x * (1.0 + torch.erf(x / 2 ** 0.5))).bfloat16().double().reshape(P, H, W, 2, 2, 2, 2, 32)   # [nb, y, x, ky, kx, ky2, kx2, c]
    full = gel.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(P, 4 * H, 4 * W, 32)
    want = torch.einsum("pmc,pyxc->pmyx", hyper.double(), full)
    assert float((got.double() - want).abs().max()) < 3e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [4096 * 3, 4096, 1000])
def test_fused_first_upscaling_layernorm_gelu_equals_the_separate_launches(rows):
    """ops.up1_ln_gelu (first transposed convolution as Linear 256 -> 4 x 64, LayerNorm2d over each tap's 64 channels, GELU: one kernel, the fp32
    convolution output never written) against ops.gemm(out_f32) + ops.norm(act=GELU) on the same operands."""
    from ullsam_amd import ops
    g = torch.Generator(device=DEV); g.manual_seed(rows)
    src = torch.randn(rows, 256, device=DEV, generator=g).bfloat16()
    w0 = (torch.randn(256, 256, device=DEV, generator=g) * 0.08).bfloat16()
    b0 = torch.randn(256, device=DEV, generator=g) * 0.1
    lw = 1.0 + 0.1 * torch.randn(64, device=DEV, generator=g)
    lb = 0.1 * torch.randn(64, device=DEV, generator=g)
    ref = ops.norm(ops.gemm(src, w0, b0, out_f32=True).reshape(rows * 4, 64), lw, lb, 1e-6, torch.bfloat16, act=ops.ACT_GELU).float()
    got = ops.up1_ln_gelu(src, w0, b0, lw, lb, 1e-6).float()
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (rows * 4, 64) and torch.isfinite(got).all()
    d = (got - ref).abs()
    assert float(d.max()) < 2e-2 * max(1.0, float(ref.abs().max())) and float(d.mean()) < 3e-4, (float(d.max()), float(d.mean()))
    x = (src.double() @ w0.double().T + b0.double()).reshape(rows * 4, 64)
    xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-6) * lw.double() + lb.double()
    want = 0.5 * xn * (1.0 + torch.erf(xn / 2 ** 0.5))
    assert float((got.double() - want).abs().max()) < 2e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("P,T,shared", [(64, 7, True), (4, 7, False), (3, 16, False), (1, 6, False), (17, 9, True)])
def test_fused_token_side_of_the_two_way_blocks_equals_the_separate_launches(P, T, shared):
    """transformer.FUSED_TOK (default for bf16): the token side of a two-way block -- self attention with its four projections, norm1, the token -> image q
    projection | the token -> image out projection, norm2, the MLP, norm3, the image -> token k / v projections -- as TWO launches around the token -> image
    attention (csrc/dectok.hip: one workgroup per prompt, bf16 MFMA with fp32 accumulation, every activation as TWO bf16 terms = ~17 bits) against the ~18 fp32 launches it replaces, through TwoWayTransformer.forward_tokens on random weights.  The AMG batch shape (64 prompts on one
    shared image), the bench's (4 images, one prompt each), the largest token count, one prompt, a prompt count that is not a multiple of anything."""
    from ullsam_amd.modeling import transformer as TR
    torch.manual_seed(P * 131 + T)
    tw = TR.TwoWayTransformer(depth=2, embedding_dim=256, num_heads=8, mlp_dim=2048).to(DEV)
    for prm in tw.parameters():
        torch.nn.init.normal_(prm, std=0.06)
    for n, prm in tw.named_parameters():
        if "norm" in n and n.endswith("weight"):
            torch.nn.init.normal_(prm, mean=1.0, std=0.1)
    tw = tw.to(torch.bfloat16)
    N = 4096
    keys = torch.randn((1 if shared else P, N, 256), device=DEV)
    key_pe = torch.randn((N, 256), device=DEV)
    tokens = torch.randn((P, T, 256), device=DEV)
    outs = {}
    old = TR.FUSED_TOK
    try:
        for on in (True, False):
            TR.FUSED_TOK = on
            q, k = tw.forward_tokens(keys, key_pe, tokens, keys_in_compute_dtype=True)
            outs[on] = (q.float(), k.float())
    finally:
        TR.FUSED_TOK = old
    torch.cuda.synchronize()
    (qa, ka), (qb, kb) = outs[True], outs[False]
    assert torch.isfinite(qa).all() and torch.isfinite(ka).all()
    print(f"fused token side P={P} T={T}: queries max diff {float((qa - qb).abs().max()):.4f} (scale {float(qb.abs().max()):.2f}), mean {float((qa - qb).abs().mean()):.5f}; "
          f"keys max diff {float((ka - kb).abs().max()):.4f}, mean {float((ka - kb).abs().mean()):.5f}")
    # (measured: queries max 4e-4 ... 2.5e-3, mean 1e-4 with two bf16 terms per activation; one term -- autocast's rounding -- gave 2e-2 / 3e-3 and cost mask IoU)
    assert float((qa - qb).abs().max()) < 1e-2 * max(1.0, float(qb.abs().max())), float((qa - qb).abs().max())
    assert float((qa - qb).abs().mean()) < 5e-4 * max(1.0, float(qb.abs().mean()))
    assert float((ka - kb).abs().max()) < 4e-2 * max(1.0, float(kb.abs().max())), float((ka - kb).abs().max())      # (the keys leave in bf16: one rounding step)
    assert float((ka - kb).abs().mean()) < 5e-4 * max(1.0, float(kb.abs().mean()))
    # a record's outputs do not depend on what it is batched with: prompt 0 alone gives the bits it has inside the batch
    if not shared and P > 1:
        q1, k1 = tw.forward_tokens(keys[:1], key_pe, tokens[:1], keys_in_compute_dtype=True)
        assert torch.equal(q1.float()[0], qa[0]) and torch.equal(k1.float()[0], ka[0])


@pytest.mark.gpu
@pytest.mark.parametrize("P", [64, 4, 1, 19])
def test_fused_hypernetwork_and_iou_heads_equal_the_separate_launches(P):
    """mask_decoder.FUSED_HEADS (default for bf16): the four hypernetwork MLPs and the IoU head (mask_decoder.py:141-149,154-176; 15 linears) as one launch
    (csrc/dectok.hip dec_heads_kernel) against the 15 fp32 launches: through MaskDecoder.predict_masks_tokens on random weights."""
    from ullsam_amd.modeling import mask_decoder as MD
    from ullsam_amd.build_sam import _build_sam
    torch.manual_seed(P)
    sam = _build_sam(128, 2, 2, [1]).to(DEV)
    md = sam.mask_decoder
    for prm in md.parameters():
        torch.nn.init.normal_(prm, std=0.06)
    md = md.to(torch.bfloat16)
    img = torch.randn((1, 4096, 256), device=DEV)
    pe = torch.randn((4096, 256), device=DEV)
    sparse = torch.randn((P, 2, 256), device=DEV)
    dense = torch.randn((1, 1, 256), device=DEV)
    outs = {}
    old = MD.FUSED_HEADS
    try:
        for on in (True, False):
            MD.FUSED_HEADS = on
            m, iou = md.predict_masks_tokens(img, pe, sparse, dense, (64, 64))
            outs[on] = (m.float(), iou.float())
    finally:
        MD.FUSED_HEADS = old
    (ma, ia), (mb, ib) = outs[True], outs[False]
    assert ma.shape == mb.shape and ia.shape == ib.shape == (P, 4)
    assert float((ia - ib).abs().max()) < 3e-2 * max(1.0, float(ib.abs().max())), float((ia - ib).abs().max())
    assert float((ma - mb).abs().max()) < 3e-2 * max(1.0, float(mb.abs().max())), float((ma - mb).abs().max())
    assert float((ma - mb).abs().mean()) < 3e-3 * max(1.0, float(mb.abs().mean()))
