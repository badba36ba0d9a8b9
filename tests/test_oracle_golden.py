"""Pin the numpy oracle against vectors captured from the reference itself (oracle/gen_golden.py).

The reference ships no tests / golden vectors of its own, so these captured
outputs are the pin.  fp32 vs fp32: tolerances are accumulation-order noise only.
"""
import numpy as np
import pytest
import torch

from oracle import ullsam_oracle as O
from tests import util as U


def _close(a, b, atol, what):
    err = float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol}"


def test_vit_tiny_matches_reference():
    g = U.gold("vit_tiny")
    P = U.vit_params(U.VIT_TINY, int(g["weight_seed"]))
    x = U.rand_image((2, 3, 160, 160), int(g["input_seed"]))
    y = O.vit_encoder(x, P, **U.vit_run_cfg(U.VIT_TINY))
    assert y.shape == g["out"].shape == (2, 64, 10, 10)
    _close(y, g["out"], 2e-4, "vit_tiny output")


def relpos_interp_params(g):
    """vit_tiny's parameters with the rel-pos tables of tests/golden/vit_tiny_relpos_interp.npz: other lengths (9 for the 7 x 7 windows, 15 for the 10 x 10 grid)."""
    P = U.vit_params(U.VIT_TINY, int(g["weight_seed"]))
    hd = U.VIT_TINY["embed_dim"] // U.VIT_TINY["num_heads"]
    for i in range(U.VIT_TINY["depth"]):
        L = int(g["len_global"] if i in U.VIT_TINY["global_attn_indexes"] else g["len_window"])
        for nm in ("rel_pos_h", "rel_pos_w"):
            P[f"blocks.{i}.attn.{nm}"] = O.fill_param(f"blocks.{i}.attn.{nm}", (L, hd), int(g["table_seed"]))
    return P


def test_vit_with_interpolated_rel_pos_tables_matches_reference():
    """get_rel_pos interpolates a table whose length is not 2 * size - 1 (image_encoder.py:306-318); fixture: the reference with such tables."""
    g = U.gold("vit_tiny_relpos_interp")
    x = U.rand_image((2, 3, 160, 160), int(g["input_seed"]))
    y = O.vit_encoder(x, relpos_interp_params(g), **U.vit_run_cfg(U.VIT_TINY))
    _close(y, g["out"], 2e-4, "vit_tiny with interpolated rel-pos tables")
    _close(O.interp_linear_rows(np.arange(6, dtype=np.float32)[:, None] * 2, 6), np.arange(6, dtype=np.float32)[:, None] * 2, 0, "identity length")


def test_decoder_matches_reference():
    g = U.gold("decoder")
    seed = int(g["weight_seed"])
    P = {}
    P.update(O.fill_state(O.prompt_encoder_shapes(prefix="prompt_encoder."), seed))
    P.update(O.fill_state(O.mask_decoder_shapes(prefix="mask_decoder."), seed))
    emb, llm, mask_in = U.decoder_inputs(int(g["input_seed"]))
    pe = O.dense_pe(P, prefix="prompt_encoder.")
    _close(pe.reshape(-1)[::29], g["dense_pe_sample"], 1e-5, "dense_pe")
    pts, lbl, boxes = g["pts"], g["lbl"], g["boxes"]

    def run(tag, points, bx, msk, llm_h, multi):
        sp, de = O.prompt_encoder(P, points, bx, msk, llm_h, prefix="prompt_encoder.")
        low, iou = O.mask_decoder(P, emb, pe, sp, de, multi, prefix="mask_decoder.")
        _close(sp, g[tag + "_sparse"], 1e-5, tag + " sparse")
        _close(de.reshape(de.shape[0], -1)[:, ::61], g[tag + "_dense_sample"], 1e-4, tag + " dense")
        ref_low = g[tag + "_low"]
        got = low if low.shape[1] == 1 else low[:, :, ::3, ::3]
        _close(got, ref_low, 2e-3 * max(1.0, float(np.abs(ref_low).max()) / 10), tag + " low_res")
        _close(iou, g[tag + "_iou"], 1e-3, tag + " iou")

    run("pts_llm_single", (pts, lbl), None, None, np.repeat(llm, 3, 0), False)
    run("pts_plain_multi", (pts, lbl), None, None, None, True)
    run("pts_box_plain", (pts, lbl), boxes, None, None, False)
    run("box_mask", None, boxes, mask_in, None, True)
    run("one_pt_llm", (pts[:1, :1], lbl[:1, :1]), None, None, llm, False)


def test_llm_tiny_matches_reference():
    g = U.gold("llm_tiny")
    P = U.llm_params(U.LLM_TINY, int(g["weight_seed"]))
    emb, mask = g["emb"], g["mask"]
    # reference default position_ids = arange (modeling_internlm2.py:893-898) even with left padding
    hid, _ = O.internlm2_model(P, U.LLM_TINY, emb, mask, prefix="language_model.")
    valid = mask.astype(bool)
    _close(hid[valid], g["hidden"][valid], 2e-4, "hidden (non-pad rows)")
    logits = O.lm_head(P, hid[:, -1], "language_model.")
    _close(logits[:, ::97], g["logits_last_sample"], 2e-3, "last logits")
    assert (logits.argmax(-1) == g["logits_last_argmax"]).all()
    toks = O.greedy_generate(P, U.LLM_TINY, emb[:1, :40], None, 12)
    assert toks.tolist() == g["greedy_tokens"].tolist(), "greedy token ids must be bit-exact"


def test_vit_h_d2_matches_reference():
    """ViT-H width (16 heads x 80) at 1024^2, one windowed + one global block: the bench's attention shapes."""
    g = U.gold("vit_h_d2")
    P = U.vit_params(U.VIT_H_D2, int(g["weight_seed"]))
    x = U.rand_image((1, 3, 1024, 1024), int(g["input_seed"]))
    y = O.vit_encoder(x, P, **U.vit_run_cfg(U.VIT_H_D2))
    assert y.shape == (1, 256, 64, 64)
    _close(y.reshape(-1)[::int(g["stride"])], g["sample"], 5e-4, "vit_h_d2 sample")
    assert abs(float(y.mean()) - float(g["mean"])) < 1e-4 and abs(float(y.std()) - float(g["std"])) < 1e-4


def test_llm_7b_l1_matches_reference():
    """One InternLM2 layer at the 7B shape (4096 / 32 heads / 8 KV heads / 14336), S = 1081, left padding on sequence 1."""
    g = U.gold("llm_7b_l1")
    c = U.LLM_7B_L1
    P = U.llm_params(c, int(g["weight_seed"]))
    emb, mask = U.llm_7b_l1_inputs(int(g["input_seed"])), g["mask"]
    hid, _ = O.internlm2_model(P, c, emb, mask, prefix="language_model.")
    valid = mask.astype(bool)[:, ::23]
    _close(hid[:, ::23, ::17][valid], g["hidden_sample"][valid], 5e-4, "hidden sample (non-pad rows)")
    logits = O.lm_head(P, hid[:, -1], "language_model.")
    _close(logits[:, ::97], g["logits_last_sample"], 5e-3, "last logits")
    assert (logits.argmax(-1) == g["logits_last_argmax"]).all()


def test_rope_variants_match_reference_oracle_and_host():
    """Plain / linear-scaled / dynamic-NTK rotary tables (modeling_internlm2.py:147-229) incl. the reference modules' stateful
    cache growth (a table built for a longer sequence is reused for shorter ones): the oracle and the host-side table builder of
    the product (InternLM2Model.rope_tables, plain torch on the CPU here) against tables produced by the reference's modules."""
    from ullsam_amd.modeling.configuration_internlm2 import InternLM2Config
    from ullsam_amd.modeling.modeling_internlm2 import InternLM2Model
    g = U.gold("rope_variants")
    hd, mp, base = int(g["head_dim"]), int(g["max_pos"]), float(g["base"])
    for tag, rs in (("plain", None), ("linear", {"type": "linear", "factor": 2.0}), ("dynamic", {"type": "dynamic", "factor": 4.0})):
        cfg = InternLM2Config(vocab_size=8, hidden_size=hd, intermediate_size=8, num_hidden_layers=0, num_attention_heads=1,
                              num_key_value_heads=1, max_position_embeddings=mp, rope_theta=base, rope_scaling=rs)
        m = InternLM2Model(cfg)
        built_for = 0   # the sequence length the reference module's cache was last (re)built for
        for i, sl in enumerate(g["seq_lens"].tolist()):
            if sl > max(built_for, mp):
                built_for = sl
            cos, sin = O.rope_tables(hd, sl, base, rs, mp, seq_len=max(built_for, 1))
            _close(cos[::3, ::5], g[f"{tag}_{i}_cos"], 2e-5, f"oracle {tag} cos #{i}")
            _close(sin[::3, ::5], g[f"{tag}_{i}_sin"], 2e-5, f"oracle {tag} sin #{i}")
            c2, s2 = m.rope_tables(sl, "cpu")
            _close(c2[:sl].numpy()[::3, ::5], g[f"{tag}_{i}_cos"], 2e-5, f"host {tag} cos #{i}")
            _close(s2[:sl].numpy()[::3, ::5], g[f"{tag}_{i}_sin"], 2e-5, f"host {tag} sin #{i}")


def test_llm_tiny_bias_linear_matches_reference():
    """config.bias=True (wqkv / wo biases) + linear RoPE scaling through the oracle."""
    g = U.gold("llm_tiny_bias_linear")
    c = dict(U.LLM_TINY, rope_scaling={"type": "linear", "factor": 2.0})
    P = O.fill_state(O.internlm2_shapes(c["hidden"], c["layers"], c["heads"], c["kv_heads"], c["inter"], c["vocab"],
                                        prefix="language_model.", bias=True), int(g["weight_seed"]))
    emb = np.random.default_rng(int(g["input_seed"])).standard_normal((2, 50, 256), dtype=np.float32) * np.float32(0.5)
    hid, _ = O.internlm2_model(P, c, emb, g["mask"], prefix="language_model.")
    valid = g["mask"].astype(bool)
    _close(hid[valid], g["hidden"][valid], 2e-4, "hidden (non-pad rows)")
    assert (O.lm_head(P, hid[:, -1], "language_model.").argmax(-1) == g["logits_last_argmax"]).all()


def test_pixel_shuffle_roundtrip_and_maps():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 64, 64, 256), dtype=np.float32)
    y = O.pixel_shuffle_v2(x)
    assert y.shape == (1, 32, 32, 1024)
    # closed form used by the HIP gather kernels: out[h2,w2,(h%2)*512+(w%2)*256+c] = x[2h2+h%2, 2w2+w%2, c]
    for (h, w, c) in [(0, 0, 0), (5, 9, 17), (63, 62, 255), (10, 11, 3)]:
        assert y[0, h // 2, w // 2, (h % 2) * 512 + (w % 2) * 256 + c] == x[0, h, w, c]


@pytest.mark.slow
def test_ullsam_tiny_matches_reference():
    g = U.gold("ullsam_tiny")
    P = U.ullsam_tiny_params(int(g["weight_seed"]))
    x = U.rand_image((1, 3, 1024, 1024), int(g["input_seed"]))
    r = O.ullsam_mask_path(P, x, g["ids"], g["pts"], g["lbl"], U.vit_run_cfg(U.VIT_SMALL), U.LLM_TINY)
    _close(r["vit_embeds"].reshape(-1)[::101], g["vit_embeds_sample"], 1e-3, "vit_embeds")
    _close(r["image_embeddings"].reshape(-1)[::37], g["img_emb_sample"], 5e-4, "image_embeddings")
    _close(r["dense_feature"].reshape(-1)[::37], g["dense_feat_sample"], 2e-3, "dense feature")
    _close(r["low_res_logits"], g["low"], 1e-3 * max(1.0, float(np.abs(g["low"]).max())), "low-res logits")
    _close(r["iou_predictions"], g["iou"], 1e-3, "iou")
    ref_mask = np.unpackbits(g["mask_bits"])[:1024 * 1024].reshape(1024, 1024).astype(bool)
    iou = O.calc_iou(r["masks"][0, 0], ref_mask)
    assert 1.0 - iou < 1e-4, f"mask IoU vs reference {iou}"
    # greedy ids through the composite embedding path
    emb = O.build_inputs_embeds(P, g["ids"], r["vit_embeds"])
    toks = O.greedy_generate(P, U.LLM_TINY, emb, None, 8)
    assert toks.tolist() == g["greedy_tokens"].tolist()


def test_sam_forward_matches_reference():
    g = U.gold("sam_forward")
    seed = int(g["weight_seed"])
    P = {}
    P.update(U.vit_params(U.VIT_SMALL, seed, "image_encoder."))
    P.update(O.fill_state(O.prompt_encoder_shapes(prefix="prompt_encoder."), seed))
    P.update(O.fill_state(O.mask_decoder_shapes(prefix="mask_decoder."), seed))
    img = U.rand_image((3, 768, 1024), int(g["input_seed"]), 255.0)
    r = O.sam_forward_one(P, img, g["pts"], g["lbl"], True, U.vit_run_cfg(U.VIT_SMALL))
    _close(r["low_res_logits"], g["low"], 1e-3 * max(1.0, float(np.abs(g["low"]).max())), "low_res_logits")
    _close(r["iou_predictions"], g["iou"], 1e-3, "iou")
    # second resize to original_size (600, 800), sam.py:161
    up = O.bilinear_resize(O.bilinear_resize(r["low_res_logits"], (1024, 1024))[..., :768, :1024], (600, 800)) > 0.0
    shape = tuple(int(v) for v in g["mask_shape"])
    ref = np.unpackbits(g["mask_bits"])[:int(np.prod(shape))].reshape(shape).astype(bool)
    assert up.shape == ref.shape
    assert 1.0 - O.calc_iou(up, ref) < 1e-4


@pytest.mark.slow
def test_vit_b_full_matches_reference():
    g = U.gold("vit_b_full")
    P = U.vit_params(U.VIT_B, int(g["weight_seed"]))
    x = U.rand_image((1, 3, 1024, 1024), int(g["input_seed"]))
    y = O.vit_encoder(x, P, **U.vit_run_cfg(U.VIT_B))
    _close(y.reshape(-1)[::int(g["stride"])], g["sample"], 2e-3, "vit_b output sample")
    assert abs(float(y.mean()) - float(g["mean"])) < 1e-4 and abs(float(y.std()) - float(g["std"])) < 1e-3


def test_amg_oracle_and_host_helpers_match_reference_vectors():
    """utils/amg.py helpers: the numpy oracle (and the pure-host product helpers) reproduce the reference's outputs bit for bit."""
    from oracle import amg_oracle as A
    from ullsam_amd.utils import amg as H
    g = U.gold("amg")
    m = g["logits"]
    assert np.array_equal(A.calculate_stability_score(m, 0.0, 1.0), g["stability"], equal_nan=True)
    assert np.array_equal(A.calculate_stability_score(m.reshape(3, 4, 96, 128), 0.25, 0.5), g["stability_b"], equal_nan=True)
    b = m > 0
    assert np.array_equal(A.batched_mask_to_box(b), g["boxes"]) and np.array_equal(A.batched_mask_to_box(b.reshape(3, 4, 96, 128)), g["boxes_4d"])
    r = A.mask_to_rle(b)
    assert [len(x["counts"]) for x in r] == g["rle_lens"].tolist()
    assert np.concatenate([x["counts"] for x in r]).tolist() == g["rle_counts"].tolist()
    assert [A.area_from_rle(x) for x in r] == g["rle_area"].tolist() == [H.area_from_rle(x) for x in r]
    assert all(np.array_equal(A.rle_to_mask(x), b[i]) and np.array_equal(H.rle_to_mask(x), b[i]) for i, x in enumerate(r))
    assert np.array_equal(A.is_box_near_crop_edge(g["boxes"], [100, 50, 228, 146], [0, 0, 400, 300]), g["near_edge"])
    assert np.array_equal(A.uncrop_boxes_xyxy(g["boxes"], [100, 50, 228, 146]), g["uncrop_boxes"])
    for mod in (A, H):
        assert np.array_equal(mod.build_point_grid(5), g["grid5"])
        gl = mod.build_all_layer_point_grids(32, 2, 2)
        assert [len(x) for x in gl] == g["grid_layers"].tolist() and np.array_equal(gl[2], g["grid_l2"])
        cb, li = mod.generate_crop_boxes((1500, 2250), 2, 512 / 1500)
        assert np.array_equal(np.asarray(cb), g["crop_boxes"]) and list(li) == g["crop_layers"].tolist()
    md = H.MaskData(a=[1, 2, 3], b=np.arange(3))
    md.filter(torch.tensor([True, False, True]))
    assert md["a"] == [1, 3] and md["b"].tolist() == [0, 2]
    assert [x for x in H.batch_iterator(2, [1, 2, 3])] == [[[1, 2]], [[3]]]


def test_torch_cpu_port_of_the_heavy_stages_equals_the_numpy_oracle():
    """bench.py's cpu_baseline times oracle/torch_port.py (the oracle's ViT block and InternLM2 layer on ATen kernels = the reference's own CPU substrate):
    the port must be the oracle's arithmetic -- windowed block with padded windows, global block, an LLM layer with grouped KV heads."""
    from oracle import torch_port as TP
    rng = np.random.default_rng(0)
    D, H = 128, 2
    P = O.fill_state(O.vit_shapes(embed_dim=D, depth=2, num_heads=H, global_attn_indexes=(1,), img_size=320, window_size=7), 0)
    x = rng.standard_normal((2, 20, 20, D), dtype=np.float32)
    PT = TP.to_torch(P)
    for i, ws in ((0, 7), (1, 0)):
        _close(TP.vit_block(torch.from_numpy(x), PT, f"blocks.{i}.", H, ws, 1e-6).numpy(), O.vit_block(x, P, f"blocks.{i}.", H, ws, 1e-6), 2e-5, f"torch port, ViT block (window {ws})")
    cfg = dict(hidden=256, layers=1, heads=4, kv_heads=2, inter=512, vocab=8, rope_theta=1e6, eps=1e-5)
    PL = O.fill_state(O.internlm2_shapes(256, 1, 4, 2, 512, 8, prefix="lm."), 0)
    emb = rng.standard_normal((2, 70, 256), dtype=np.float32)
    ref, _ = O.internlm2_model(PL, cfg, emb, prefix="lm.")
    got = O.rms_norm(TP.internlm2_layer(torch.from_numpy(emb), TP.to_torch(PL), "lm.model.layers.0.", cfg).numpy(), PL["lm.model.norm.weight"], 1e-5)
    _close(got, ref, 2e-5, "torch port, InternLM2 layer")
