"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference).

Run in the build container only (the reference never travels to the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [case ...]

Each fixture stores: the generating config, the seeds (weights come from
oracle.ullsam_oracle.fill_param -- regenerable anywhere without the reference), the inputs that are
not seed-derivable, and the reference's outputs.  Fixtures are data only; no reference source is copied.
"""
from __future__ import annotations

import os
import sys
import time
from functools import partial

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import ullsam_oracle as O  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_grad_enabled(False)


def fill_module(mod: torch.nn.Module, seed: int, prefix: str = ""):
    """Overwrite every parameter/persistent buffer of `mod` with fill_param(prefix+name)."""
    sd = mod.state_dict()
    new = {k: torch.from_numpy(O.fill_param(prefix + k, tuple(v.shape), seed)).to(v.dtype) for k, v in sd.items()}
    mod.load_state_dict(new, strict=True)
    return {prefix + k: tuple(v.shape) for k, v in sd.items()}


def rand_image(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).random(shape, dtype=np.float32) * scale).astype(np.float32)


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------------------
RELPOS_INTERP_LEN = {"window": 9, "global": 15}     # table lengths of the interpolation fixture (what window 5 / a grid of 8 would have trained): the model needs 13 / 19


def case_vit_tiny_relpos_interp():
    """vit_tiny with rel_pos tables of ANOTHER length than 2 * size - 1 (a checkpoint trained at another resolution): get_rel_pos interpolates them linearly
    (image_encoder.py:306-318).  The reference module is built as usual and its tables are replaced before the forward."""
    from modeling.image_encoder import ImageEncoderViT
    cfg = dict(img_size=160, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, out_chans=64,
               qkv_bias=True, use_rel_pos=True, window_size=7, global_attn_indexes=[1])
    m = ImageEncoderViT(norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **cfg).eval()
    fill_module(m, seed=0)
    for i, blk in enumerate(m.blocks):
        L = RELPOS_INTERP_LEN["global" if i in cfg["global_attn_indexes"] else "window"]
        for nm in ("rel_pos_h", "rel_pos_w"):
            t = O.fill_param(f"blocks.{i}.attn.{nm}", (L, 64), 5)          # seed 5: these replace the seed-0 tables of the usual length
            setattr(blk.attn, nm, torch.nn.Parameter(torch.from_numpy(t)))
    x = rand_image((2, 3, 160, 160), seed=1)
    y = m(torch.from_numpy(x)).numpy()
    save("vit_tiny_relpos_interp", cfg=np.array(repr(cfg)), weight_seed=0, table_seed=5, input_seed=1, len_window=RELPOS_INTERP_LEN["window"],
         len_global=RELPOS_INTERP_LEN["global"], out=y)


def case_vit_tiny():
    from modeling.image_encoder import ImageEncoderViT
    cfg = dict(img_size=160, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, out_chans=64,
               qkv_bias=True, use_rel_pos=True, window_size=7, global_attn_indexes=[1])
    m = ImageEncoderViT(norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **cfg).eval()
    fill_module(m, seed=0)
    x = rand_image((2, 3, 160, 160), seed=1)
    y = m(torch.from_numpy(x)).numpy()
    save("vit_tiny", cfg=np.array(repr(cfg)), weight_seed=0, input_seed=1, out=y)


def case_vit_b_full():
    from build_sam import sam_model_registry
    sam = sam_model_registry["vit_b"]()
    fill_module(sam.image_encoder, seed=0)
    x = rand_image((1, 3, 1024, 1024), seed=1)
    t = time.time()
    y = sam.image_encoder(torch.from_numpy(x)).numpy()
    print(f"  reference ViT-B forward {time.time() - t:.1f}s")
    flat = y.reshape(-1)
    save("vit_b_full", weight_seed=0, input_seed=1, stride=37, sample=flat[::37].copy(),
         mean=np.float64(flat.mean()), std=np.float64(flat.std()), absmax=np.float64(np.abs(flat).max()))


def _sam_small(depth=2, embed_dim=128, heads=2, glob=(1,)):
    from modeling import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer
    sam = Sam(
        image_encoder=ImageEncoderViT(depth=depth, embed_dim=embed_dim, img_size=1024, mlp_ratio=4,
                                      norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=heads, patch_size=16,
                                      qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(glob), window_size=14,
                                      out_chans=256),
        prompt_encoder=PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024),
                                     mask_in_chans=16),
        mask_decoder=MaskDecoder(num_multimask_outputs=3,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                                 transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256),
    ).eval()
    return sam


def case_decoder():
    """Prompt encoder + mask decoder at full size on a synthetic (LayerNorm-like) image embedding."""
    sam = _sam_small()
    fill_module(sam.prompt_encoder, seed=0, prefix="prompt_encoder.")
    fill_module(sam.mask_decoder, seed=0, prefix="mask_decoder.")
    rng = np.random.default_rng(2)
    emb = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    llm = rng.standard_normal((1, 256, 64, 64), dtype=np.float32) * 3.0 + 0.5
    pts = np.array([[[300.0, 400.0], [700.0, 120.0]], [[512.0, 512.0], [10.0, 1000.0]], [[5.0, 5.0], [900.0, 900.0]]], np.float32)
    lbl = np.array([[1, 0], [1, 1], [0, -1]], np.int32)
    boxes = np.array([[100.0, 150.0, 600.0, 700.0], [0.0, 0.0, 1023.0, 1023.0], [400.0, 300.0, 500.0, 350.0]], np.float32)
    mask_in = rng.standard_normal((3, 1, 256, 256), dtype=np.float32)
    pe_t = sam.prompt_encoder.get_dense_pe()
    # emb / llm / mask_in are regenerable: decoder_inputs(seed=2) in tests/util.py draws them in this order
    out = {"input_seed": 2, "pts": pts, "lbl": lbl, "boxes": boxes, "dense_pe_sample": pe_t.numpy().reshape(-1)[::29].copy()}

    def run(tag, points, bx, msk, llm_h, multi):
        sp, de = sam.prompt_encoder(points=points, boxes=bx, masks=msk, llm_hidden_states=llm_h)
        low, iou = sam.mask_decoder(image_embeddings=torch.from_numpy(emb), image_pe=pe_t, sparse_prompt_embeddings=sp,
                                    dense_prompt_embeddings=de, multimask_output=multi)
        out[tag + "_sparse"] = sp.numpy()
        out[tag + "_dense_sample"] = de.numpy().reshape(de.shape[0], -1)[:, ::61].copy()
        out[tag + "_low"] = low.numpy() if low.shape[1] == 1 else low.numpy()[:, :, ::3, ::3].copy()
        out[tag + "_iou"] = iou.numpy()

    P = (torch.from_numpy(pts), torch.from_numpy(lbl))
    llm3 = torch.from_numpy(llm).repeat(3, 1, 1, 1)
    run("pts_llm_single", P, None, None, llm3, False)          # the uLLSAM metric path (app.py:617-631)
    run("pts_plain_multi", P, None, None, None, True)          # plain SAM, multimask
    run("pts_box_plain", P, torch.from_numpy(boxes), None, None, False)
    run("box_mask", None, torch.from_numpy(boxes), torch.from_numpy(mask_in), None, True)
    P1 = (torch.from_numpy(pts[:1, :1]), torch.from_numpy(lbl[:1, :1]))
    run("one_pt_llm", P1, None, None, torch.from_numpy(llm), False)
    save("decoder", weight_seed=0, **out)


LLM_TINY = dict(architectures=["InternLM2ForCausalLM"], vocab_size=92553, hidden_size=256, intermediate_size=512,
                num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, bias=False,
                max_position_embeddings=32768, rope_theta=1000000, rms_norm_eps=1e-5, attn_implementation="eager")


def _tiny_llm():
    from modeling.configuration_internlm2 import InternLM2Config
    from modeling.modeling_internlm2 import InternLM2ForCausalLM
    cfg = InternLM2Config(**LLM_TINY)
    cfg.rope_scaling = None  # transformers>=5 aliases it to a dict (SURVEY.md 8(c) workaround 1)
    m = InternLM2ForCausalLM(cfg).eval()
    return m


def _ref_greedy(lm, emb, mask, max_new, eos=92542):
    """Manual greedy loop over the reference's own forward + tuple KV cache (SURVEY.md 8(c) workaround 2)."""
    pos = mask.long().cumsum(-1) - 1
    pos.masked_fill_(mask == 0, 1)
    o = lm(inputs_embeds=emb, attention_mask=mask, position_ids=pos, use_cache=True, return_dict=True)
    past = o.past_key_values
    toks = []
    logits0 = o.logits[:, -1].numpy().copy()
    for _ in range(max_new):
        t = int(o.logits[0, -1].argmax())
        toks.append(t)
        if t == eos:
            break
        mask = torch.cat([mask, torch.ones((1, 1), dtype=mask.dtype)], 1)
        pos = (mask.long().cumsum(-1) - 1)[:, -1:]
        o = lm(input_ids=torch.tensor([[t]]), attention_mask=mask, position_ids=pos, past_key_values=past,
               use_cache=True, return_dict=True)
        past = o.past_key_values
    return np.asarray(toks, np.int64), logits0


def case_llm_tiny():
    lm = _tiny_llm()
    fill_module(lm, seed=0, prefix="language_model.")
    rng = np.random.default_rng(3)
    B, S = 2, 70
    emb = (rng.standard_normal((B, S, 256), dtype=np.float32) * 0.5)
    mask = np.ones((B, S), np.int64)
    mask[1, :9] = 0  # left padding on sample 1
    o = lm(inputs_embeds=torch.from_numpy(emb), attention_mask=torch.from_numpy(mask), use_cache=False,
           output_hidden_states=True, return_dict=True)
    hidden = o.hidden_states[-1].numpy()
    logits_last = o.logits[:, -1].numpy()
    # greedy (B=1, no padding), prompt = first 40 positions of sample 0
    toks, logits0 = _ref_greedy(lm, torch.from_numpy(emb[:1, :40]), torch.ones((1, 40), dtype=torch.long), 12)
    save("llm_tiny", weight_seed=0, cfg=np.array(repr(LLM_TINY)), emb=emb, mask=mask, hidden=hidden,
         logits_last_sample=logits_last[:, ::97].copy(), logits_last_argmax=logits_last.argmax(-1),
         greedy_tokens=toks, greedy_logits0_sample=logits0[:, ::97].copy())


def case_ullsam_tiny():
    """Composite InternVLSAMModel: shallow ViT (1024^2 in, 256x64x64 out) + tiny LLM + full decoder."""
    from modeling.configuration_internvl_chat import InternVLChatConfig
    from modeling.modeling_internvl_sam import InternVLSAMModel
    sam = _sam_small()
    cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT-B-16"]}, llm_config=dict(LLM_TINY),
                             downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
    cfg.llm_config.rope_scaling = None
    m = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder,
                         mask_decoder=sam.mask_decoder).eval()
    fill_module(m, seed=0)
    x = rand_image((1, 3, 1024, 1024), seed=1)
    ids = O.make_input_ids(n_text_pre=20, n_text_post=34, seed=1)  # S = 1+20+1+1024+1+34 = 1081
    assert ids.shape[1] == 1081
    tids = torch.from_numpy(ids)
    out = m(pixel_values=torch.from_numpy(x), input_ids=tids, attention_mask=torch.ones_like(tids),
            image_flags=(tids == 92546)[..., None].long(), return_dict=True, use_cache=False, output_hidden_states=True)
    dense_feat = out.hidden_states.numpy()          # [1,256,64,64]
    img_emb = out.image_embeddings.numpy()
    pts = np.array([[[512.0, 384.0], [100.0, 900.0]]], np.float32)
    lbl = np.array([[1, 0]], np.int32)
    sp, de = m.prompt_encoder(points=(torch.from_numpy(pts), torch.from_numpy(lbl)), boxes=None, masks=None,
                              llm_hidden_states=out.hidden_states)
    low, iou = m.mask_decoder(image_embeddings=out.image_embeddings, image_pe=m.prompt_encoder.get_dense_pe(),
                              sparse_prompt_embeddings=sp, dense_prompt_embeddings=de, multimask_output=False)
    up = torch.nn.functional.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)
    mask = (up[0, 0].sigmoid() > 0.5).numpy()
    # greedy tokens through the composite's embedding path (generate() :407-431 semantics)
    vit_embeds, _ = m.extract_feature(torch.from_numpy(x))
    emb = m.language_model.get_input_embeddings()(tids).clone()
    emb[tids == 92546] = vit_embeds.reshape(-1, emb.shape[-1])
    toks, logits0 = _ref_greedy(m.language_model, emb, torch.ones_like(tids), 8)
    save("ullsam_tiny", weight_seed=0, input_seed=1, ids=ids, pts=pts, lbl=lbl,
         vit_embeds_sample=vit_embeds.numpy().reshape(-1)[::101].copy(),
         img_emb_sample=img_emb.reshape(-1)[::37].copy(), dense_feat_sample=dense_feat.reshape(-1)[::37].copy(),
         logits_last_sample=out.logits[0, -1].numpy()[::97].copy(),
         low=low.numpy(), iou=iou.numpy(), up_sample=up.numpy().reshape(-1)[::53].copy(),
         mask_bits=np.packbits(mask), greedy_tokens=toks, greedy_logits0_sample=logits0[:, ::97].copy())


def case_sam_forward():
    """Sam.forward (sam.py:53-131) on a non-square image: preprocess, per-image loop, postprocess."""
    sam = _sam_small()
    fill_module(sam, seed=0)
    img = rand_image((3, 768, 1024), seed=4, scale=255.0)
    pts = np.array([[[500.0, 375.0]], [[200.0, 600.0]]], np.float32)
    lbl = np.array([[1], [1]], np.int32)
    out = sam([{"image": torch.from_numpy(img), "original_size": (600, 800), "point_coords": torch.from_numpy(pts),
                "point_labels": torch.from_numpy(lbl)}], multimask_output=True)[0]
    save("sam_forward", weight_seed=0, input_seed=4, pts=pts, lbl=lbl, low=out["low_res_logits"].numpy(),
         iou=out["iou_predictions"].numpy(), mask_bits=np.packbits(out["masks"].numpy()),
         mask_shape=np.array(out["masks"].shape))


def case_amg():
    """Helper functions of utils/amg.py run individually on seeded inputs (the reference has no generator class)."""
    from utils import amg as R
    rng = np.random.default_rng(6)
    out = {}
    # logits with blobs so masks have structure; a few degenerate ones (all below / all above threshold)
    yy, xx = np.mgrid[0:96, 0:128].astype(np.float32)
    masks = np.stack([4.0 * np.exp(-(((xx - rng.uniform(0, 128)) / rng.uniform(5, 40)) ** 2 + ((yy - rng.uniform(0, 96)) / rng.uniform(5, 40)) ** 2))
                      - 1.5 + 0.8 * rng.standard_normal((96, 128)).astype(np.float32) for _ in range(12)]).astype(np.float32)
    masks[3] = -5.0
    masks[7] = 5.0
    out["logits"] = masks
    t = torch.from_numpy(masks)
    out["stability"] = R.calculate_stability_score(t, 0.0, 1.0).numpy()
    out["stability_b"] = R.calculate_stability_score(t.reshape(3, 4, 96, 128), 0.25, 0.5).numpy()
    binm = t > 0.0
    out["boxes"] = R.batched_mask_to_box(binm).numpy()
    out["boxes_4d"] = R.batched_mask_to_box(binm.reshape(3, 4, 96, 128)).numpy()
    rles = R.mask_to_rle_pytorch(binm)
    out["rle_lens"] = np.array([len(r["counts"]) for r in rles])
    out["rle_counts"] = np.concatenate([np.asarray(r["counts"], np.int64) for r in rles])
    out["rle_area"] = np.array([R.area_from_rle(r) for r in rles])
    assert all((R.rle_to_mask(r) == binm[i].numpy()).all() for i, r in enumerate(rles))
    crop, orig = [100, 50, 228, 146], [0, 0, 400, 300]
    out["near_edge"] = R.is_box_near_crop_edge(torch.from_numpy(out["boxes"]), crop, orig).numpy()
    out["near_edge_full"] = R.is_box_near_crop_edge(torch.from_numpy(out["boxes"]), [0, 0, 128, 96], [0, 0, 128, 96]).numpy()
    out["uncrop_boxes"] = R.uncrop_boxes_xyxy(torch.from_numpy(out["boxes"]), crop).numpy()
    out["uncrop_masks_sum"] = R.uncrop_masks(binm, crop, 300, 400).sum((-1, -2)).numpy()
    out["grid5"] = R.build_point_grid(5)
    gl = R.build_all_layer_point_grids(32, 2, 2)
    out["grid_layers"] = np.array([len(g) for g in gl])
    out["grid_l2"] = gl[2]
    cb, li = R.generate_crop_boxes((1500, 2250), 2, 512 / 1500)
    out["crop_boxes"] = np.asarray(cb)
    out["crop_layers"] = np.asarray(li)
    out["xywh"] = np.asarray(R.box_xyxy_to_xywh(torch.tensor([10, 20, 50, 80])))
    save("amg", input_seed=6, **out)


def case_chat_prompt():
    """Prompt strings the reference's chat() builds (modeling_internvl_sam.py:272-335 over conversation.py's "internlm2-chat"
    template) for a first turn, a turn with history and a text-only turn: pins the host loop's query construction."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("ref_conversation", "/root/reference/modeling/conversation.py")
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    cases = []
    for system, history, question, patches, n_img_tok in (
            ("SYS", None, "Describe the cells.", [1], 3),
            ("You are a microscopy assistant.", [("first?\n<image>", "an answer")], "and now?", [1], 2),
            ("SYS", None, "<image>\n<image>\ncompare", [1, 2], 2),
            ("SYS", None, "no picture here", [], 4)):
        q = question
        has_image = len(patches) > 0
        if history is None and has_image and "<image>" not in q:
            q = q + "\n<image>"
        t = conv.get_conv_template("internlm2-chat")
        t.system_message = system
        for (oq, oa) in (history or []):
            t.append_message(t.roles[0], oq)
            t.append_message(t.roles[1], oa)
        t.append_message(t.roles[0], q)
        t.append_message(t.roles[1], None)
        query = t.get_prompt()
        for n in patches:
            query = query.replace("<image>", "<img>" + "<IMG_CONTEXT>" * n_img_tok * n + "</img>", 1)
        cases.append({"system": system, "history": history, "question": question, "num_patches_list": patches, "num_image_token": n_img_tok,
                      "query": query, "sep": t.sep.strip()})
    json.dump(cases, open(os.path.join(OUT, "chat_prompt.json"), "w"), ensure_ascii=False, indent=1)


def case_vit_h_d2():
    """ViT-H-width encoder (D=1280, 16 heads, head_dim 80: build_sam.py:14-21) cut to depth 2 = one windowed (14x14, 25 padded
    windows per image) + one global (64x64 grid) block at 1024^2 -- the attention-kernel instantiations the bench runs.  Also the
    reference's own bf16 behaviour: the same module under torch.autocast("cpu", bfloat16), to bound our bf16 mode against it."""
    from modeling.image_encoder import ImageEncoderViT
    cfg = dict(img_size=1024, patch_size=16, embed_dim=1280, depth=2, num_heads=16, mlp_ratio=4, out_chans=256,
               qkv_bias=True, use_rel_pos=True, window_size=14, global_attn_indexes=[1])
    m = ImageEncoderViT(norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **cfg).eval()
    fill_module(m, seed=0)
    x = torch.from_numpy(rand_image((1, 3, 1024, 1024), seed=1))
    t = time.time()
    y = m(x).numpy()
    print(f"  reference ViT-H/2 forward {time.time() - t:.1f}s")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        yb = m(x).float().numpy()
    flat, fb = y.reshape(-1), yb.reshape(-1)
    save("vit_h_d2", cfg=np.array(repr(cfg)), weight_seed=0, input_seed=1, stride=37, sample=flat[::37].copy(),
         mean=np.float64(flat.mean()), std=np.float64(flat.std()), absmax=np.float64(np.abs(flat).max()),
         autocast_bf16_max_err=np.float64(np.abs(fb - flat).max()), autocast_bf16_mean_err=np.float64(np.abs(fb - flat).mean()))


LLM_7B_L1 = dict(architectures=["InternLM2ForCausalLM"], vocab_size=92553, hidden_size=4096, intermediate_size=14336,
                 num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8, bias=False,
                 max_position_embeddings=32768, rope_theta=1000000, rms_norm_eps=1e-5, attn_implementation="eager")


def case_llm_7b_l1():
    """One InternLM2 layer at the 7B shape the bench runs (hidden 4096, 32 heads / 8 KV heads, intermediate 14336), S = 1081,
    batch 2 with left padding on the second sequence (modeling_internlm2.py:854-984)."""
    from modeling.configuration_internlm2 import InternLM2Config
    from modeling.modeling_internlm2 import InternLM2ForCausalLM
    cfg = InternLM2Config(**LLM_7B_L1)
    cfg.rope_scaling = None
    lm = InternLM2ForCausalLM(cfg).eval()
    fill_module(lm, seed=0, prefix="language_model.")
    rng = np.random.default_rng(8)
    B, S = 2, 1081
    emb = rng.standard_normal((B, S, 4096), dtype=np.float32) * 0.5   # regenerable: default_rng(8), this draw
    mask = np.ones((B, S), np.int64)
    mask[1, :13] = 0
    t = time.time()
    o = lm(inputs_embeds=torch.from_numpy(emb), attention_mask=torch.from_numpy(mask), use_cache=False,
           output_hidden_states=True, return_dict=True)
    print(f"  reference 7B-shaped layer forward {time.time() - t:.1f}s")
    hidden = o.hidden_states[-1].numpy()            # post-final-norm [B,S,4096]
    logits_last = o.logits[:, -1].numpy()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ob = lm(inputs_embeds=torch.from_numpy(emb), attention_mask=torch.from_numpy(mask), use_cache=False,
                output_hidden_states=True, return_dict=True)
    hb = ob.hidden_states[-1].float().numpy()
    valid = mask.astype(bool)
    save("llm_7b_l1", weight_seed=0, input_seed=8, cfg=np.array(repr(LLM_7B_L1)), mask=mask,
         hidden_sample=hidden[:, ::23, ::17].copy(), hidden_absmax=np.float64(np.abs(hidden[valid]).max()),
         logits_last_sample=logits_last[:, ::97].copy(), logits_last_argmax=logits_last.argmax(-1),
         autocast_bf16_max_err=np.float64(np.abs(hb - hidden)[valid].max()),
         autocast_bf16_mean_err=np.float64(np.abs(hb - hidden)[valid].mean()))


def case_rope_variants():
    """cos/sin caches of the reference's three rotary modules (modeling_internlm2.py:147-229) incl. their stateful growth:
    each entry = the table returned for a sequence of forward(seq_len) calls on ONE module instance."""
    from modeling import modeling_internlm2 as M
    out = {}
    x = torch.zeros(1, 1, 1, 128)
    for tag, cls, kw in (("plain", M.InternLM2RotaryEmbedding, {}),
                         ("linear", M.InternLM2LinearScalingRotaryEmbedding, {"scaling_factor": 2.0}),
                         ("dynamic", M.InternLM2DynamicNTKScalingRotaryEmbedding, {"scaling_factor": 4.0})):
        m = cls(128, max_position_embeddings=64, base=1000000, **kw)
        for i, sl in enumerate((40, 100, 70)):   # below max_pos, beyond it (dynamic: base rescaled from 100), then shorter again
            cos, sin = m(x, seq_len=sl)
            out[f"{tag}_{i}_cos"] = cos.numpy()[::3, ::5].copy()
            out[f"{tag}_{i}_sin"] = sin.numpy()[::3, ::5].copy()
    save("rope_variants", head_dim=128, max_pos=64, base=1000000.0, seq_lens=np.array([40, 100, 70]), **out)


def case_llm_tiny_bias_linear():
    """Tiny InternLM2 with config.bias=True (wqkv / wo biases, modeling_internlm2.py:300-308) and linear RoPE scaling (:184-200)."""
    from modeling.configuration_internlm2 import InternLM2Config
    from modeling.modeling_internlm2 import InternLM2ForCausalLM
    c = dict(LLM_TINY)
    c["bias"] = True
    cfg = InternLM2Config(**c)
    cfg.rope_scaling = {"type": "linear", "factor": 2.0}
    lm = InternLM2ForCausalLM(cfg).eval()
    assert type(lm.model.layers[0].attention.rotary_emb).__name__ == "InternLM2LinearScalingRotaryEmbedding"
    fill_module(lm, seed=0, prefix="language_model.")
    rng = np.random.default_rng(9)
    emb = rng.standard_normal((2, 50, 256), dtype=np.float32) * 0.5
    mask = np.ones((2, 50), np.int64)
    mask[0, :6] = 0
    o = lm(inputs_embeds=torch.from_numpy(emb), attention_mask=torch.from_numpy(mask), use_cache=False,
           output_hidden_states=True, return_dict=True)
    save("llm_tiny_bias_linear", weight_seed=0, input_seed=9, mask=mask, hidden=o.hidden_states[-1].numpy(),
         logits_last_argmax=o.logits[:, -1].numpy().argmax(-1))



VIT_H_FULL = dict(img_size=1024, patch_size=16, embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4, out_chans=256,
                  qkv_bias=True, use_rel_pos=True, window_size=14, global_attn_indexes=[7, 15, 23, 31])   # build_sam.py:14-21
LLM_7B_FULL = dict(LLM_7B_L1, num_hidden_layers=32)
FULL_STAGES_VIT = (7, 15, 23, 31)     # block indices whose OUTPUT is sampled (after 8 / 16 / 24 / 32 blocks)
FULL_STAGES_LLM = (7, 15, 23, 31)     # decoder layers whose OUTPUT is sampled
FULL_STRIDE = 997


def fill_module_inplace(mod: torch.nn.Module, seed: int, prefix: str = "", workers: int = 8):
    """fill_module without a second copy of the weights (the 7B-shaped model is 31 GB in fp32): parameters and persistent buffers are
    overwritten one by one, generated on a thread pool (numpy's generators release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    sd = mod.state_dict()

    def one(kv):
        k, v = kv
        v.copy_(torch.from_numpy(O.fill_param(prefix + k, tuple(v.shape), seed)).to(v.dtype))

    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(one, sd.items()))


def _build_full_depth():
    """The bench configuration as the reference builds it (build_sam.py:14-21 + train_joint_v2.py:1424-1461), parameters left
    uninitialised (31 GB of kaiming draws would take minutes and are overwritten anyway)."""
    from modeling import ImageEncoderViT
    from modeling.configuration_internvl_chat import InternVLChatConfig
    from modeling.modeling_internvl_sam import InternVLSAMModel
    saved = (torch.nn.Linear.reset_parameters, torch.nn.Embedding.reset_parameters, torch.nn.init.normal_, torch.nn.init.trunc_normal_)
    torch.nn.Linear.reset_parameters = lambda self: None
    torch.nn.Embedding.reset_parameters = lambda self: None
    torch.nn.init.normal_ = lambda t, *a, **k: t
    torch.nn.init.trunc_normal_ = lambda t, *a, **k: t
    try:
        sam = _sam_small()
        vit = ImageEncoderViT(norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **VIT_H_FULL)
        cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT-H-16"]}, llm_config=dict(LLM_7B_FULL),
                                 downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
        cfg.llm_config.rope_scaling = None
        m = InternVLSAMModel(cfg, vision_model=vit, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder).eval()
    finally:
        torch.nn.Linear.reset_parameters, torch.nn.Embedding.reset_parameters, torch.nn.init.normal_, torch.nn.init.trunc_normal_ = saved
    return m


FULL_TILE_SEEDS = (3, 5, 27, 32)   # fp32 mask fill 0.951 / 0.528 / 0.481 / 0.597 (tools/probes/fill_scan.py): three of the four masks cut through the middle of the logits' distribution


def case_full_depth():
    """The bench configuration at FULL depth through the reference (app.py:580-645 call sequence): ViT-H x 32 blocks + a 7B-shaped
    InternLM2 x 32 layers at S = 1081 + prompt encoder + mask decoder + x4 upsample, on FOUR synthetic microscopy tiles
    (ullsam_amd/utils/synthetic.py; the tiles bench.py times), in fp32 and under torch.autocast("cpu", bfloat16).  Per tile it stores strided
    samples after ViT block 8 / 16 / 24 / 32 and LLM layer 8 / 16 / 24 / 32, the image embedding, the dense feature, the low-res logits, the
    mask, and the reference's own autocast-vs-fp32 error at every one of those stages (the bound our bf16 mode is held to).  Keys carry the
    tile's index: `low_0`, `vit8_2_ac_mean_err`, ...; `tile_seeds` lists the tiles."""
    from ullsam_amd.utils.synthetic import microscopy_batch
    seeds = tuple(int(v) for v in os.environ.get("FULL_DEPTH_TILE_SEEDS", ",".join(map(str, FULL_TILE_SEEDS))).split(","))
    t = time.time()
    m = _build_full_depth()
    fill_module_inplace(m, seed=0)
    print(f"  built + filled in {time.time() - t:.0f}s")
    lbl = np.array([[1]], np.int32)
    ids = O.make_input_ids(n_text_pre=20, n_text_post=34, seed=1)
    assert ids.shape[1] == 1081
    tids = torch.from_numpy(ids)

    def run(x, pts):
        st = {}
        hooks = []
        for i in FULL_STAGES_VIT:
            hooks.append(m.vision_model.blocks[i].register_forward_hook(
                lambda mod, inp, out, i=i: st.__setitem__(f"vit{i + 1}", out.detach().float().numpy().reshape(-1)[::FULL_STRIDE].copy())))
        for i in FULL_STAGES_LLM:
            hooks.append(m.language_model.model.layers[i].register_forward_hook(
                lambda mod, inp, out, i=i: st.__setitem__(f"llm{i + 1}", out[0].detach().float().numpy().reshape(-1)[::FULL_STRIDE].copy())))
        out = m(pixel_values=x, input_ids=tids, attention_mask=torch.ones_like(tids), image_flags=(tids == 92546)[..., None].long(),
                return_dict=True, use_cache=False, output_hidden_states=True)
        for h in hooks:
            h.remove()
        sp, de = m.prompt_encoder(points=(torch.from_numpy(pts), torch.from_numpy(lbl)), boxes=None, masks=None,
                                  llm_hidden_states=out.hidden_states)
        low, iou = m.mask_decoder(image_embeddings=out.image_embeddings, image_pe=m.prompt_encoder.get_dense_pe(),
                                  sparse_prompt_embeddings=sp, dense_prompt_embeddings=de, multimask_output=False)
        up = torch.nn.functional.interpolate(low.float(), (1024, 1024), mode="bilinear", align_corners=False)
        st["img_emb"] = out.image_embeddings.float().numpy().reshape(-1)[::37].copy()
        st["dense_feat"] = out.hidden_states.float().numpy().reshape(-1)[::37].copy()
        st["low"] = low.float().numpy()
        st["iou_pred"] = iou.float().numpy()
        st["mask"] = (up[0, 0].sigmoid() > 0.5).numpy()     # app.py:640-645
        return st

    out = {}
    for ti, seed in enumerate(seeds):
        x_np, pts = microscopy_batch([seed])
        x = torch.from_numpy(x_np)
        t = time.time()
        f = run(x, pts)
        t32 = time.time() - t
        t = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16, cache_enabled=False):
            b = run(x, pts)
        print(f"  tile {seed}: reference fp32 forward {t32:.0f}s, autocast-bf16 forward {time.time() - t:.0f}s")
        print("  stage            mean|x|    autocast mean|d|  max|d|   rel")
        for k in [f"vit{i + 1}" for i in FULL_STAGES_VIT] + ["img_emb"] + [f"llm{i + 1}" for i in FULL_STAGES_LLM] + ["dense_feat", "low"]:
            d = np.abs(b[k].astype(np.float64) - f[k])
            out[f"{k}_{ti}"] = f[k]
            out[f"{k}_{ti}_ac_mean_err"] = np.float64(d.mean())
            out[f"{k}_{ti}_ac_max_err"] = np.float64(d.max())
            out[f"{k}_{ti}_mean_abs"] = np.float64(np.abs(f[k]).mean())
            print(f"  {k:14s} {np.abs(f[k]).mean():10.4f} {d.mean():14.5f} {d.max():10.4f} {d.mean() / np.abs(f[k]).mean():8.4f}")
        ac_iou = O.calc_iou(b["mask"], f["mask"])
        lo = f["low"].reshape(-1)
        print(f"  mask fill {f['mask'].mean():.4f}; autocast mask IoU vs fp32 {ac_iou:.6f}; low-res logits mean|x| {np.abs(lo).mean():.3f}, "
              f"share within the autocast mean error of 0: {(np.abs(lo) < out[f'low_{ti}_ac_mean_err']).mean():.5f}; "
              f"percentiles {np.percentile(lo, [1, 10, 25, 50, 75, 90, 99]).round(2)}", flush=True)
        out[f"pts_{ti}"] = pts
        out[f"mask_bits_{ti}"] = np.packbits(f["mask"])
        out[f"mask_fill_{ti}"] = np.float64(f["mask"].mean())
        out[f"iou_pred_{ti}"] = f["iou_pred"]
        out[f"ac_mask_iou_{ti}"] = np.float64(ac_iou)
        out[f"ac_iou_pred_{ti}"] = b["iou_pred"]
    save("full_depth", weight_seed=0, tile_seeds=np.asarray(seeds, np.int64), ids_seed=1, stride=FULL_STRIDE, lbl=lbl, **out)

DECODE_7B_NEW = 16
DECODE_7B_PAD = (0, 0, 11, 0)      # left-padding positions per prompt of the batch (one prompt is shorter and left-padded)


def decode_7b_ids():
    """-> (ids [4, 1081], mask [4, 1081]): four synthetic prompts around 1024 image tokens each; prompt 2 has 11 fewer text tokens
    in front and is LEFT-padded with id 2 under attention_mask 0 (what the tokenizer's padding_side='left' produces, modeling_internvl_sam.py:356-360)."""
    rows, masks = [], []
    for b, pad in enumerate(DECODE_7B_PAD):
        r = O.make_input_ids(n_text_pre=20 - pad, n_text_post=34, seed=40 + b)[0]
        rows.append(np.concatenate([np.full(pad, 2, np.int64), r]))
        masks.append(np.concatenate([np.zeros(pad, np.int64), np.ones(r.size, np.int64)]))
    ids, mask = np.stack(rows), np.stack(masks)
    assert ids.shape == (4, 1081)
    return ids, mask


def _ref_greedy_batched(lm, emb, mask, max_new):
    """Greedy decoding of a BATCH through the reference's own forward and tuple KV cache, the loop HF's generate ran around
    prepare_inputs_for_generation (modeling_internlm2.py:1112-1149: first step inputs_embeds, later steps the last id,
    position_ids = cumsum(mask) - 1 with 1 on padding).  No eos stop: every step's raw argmax is recorded, with the second choice and
    the top-2 logit margin (consumers cut a row at its first eos)."""
    B = emb.shape[0]
    pos = mask.long().cumsum(-1) - 1
    pos.masked_fill_(mask == 0, 1)
    o = lm(inputs_embeds=emb, attention_mask=mask, position_ids=pos, use_cache=True, return_dict=True)
    ids = np.zeros((B, max_new), np.int64)
    second = np.zeros((B, max_new), np.int64)
    margin = np.zeros((B, max_new), np.float32)
    top1 = np.zeros((B, max_new), np.float32)
    logits0 = o.logits[:, -1].float().numpy().copy()
    for s in range(max_new):
        last = o.logits[:, -1].float()
        v, i = last.topk(2, -1)
        ids[:, s], second[:, s] = i[:, 0].numpy(), i[:, 1].numpy()
        margin[:, s], top1[:, s] = (v[:, 0] - v[:, 1]).numpy(), v[:, 0].numpy()
        if s + 1 == max_new:
            break
        past = o.past_key_values
        mask = torch.cat([mask, torch.ones((B, 1), dtype=mask.dtype)], 1)
        pos = (mask.long().cumsum(-1) - 1)[:, -1:]
        o = lm(input_ids=i[:, :1].contiguous(), attention_mask=mask, position_ids=pos, past_key_values=past, use_cache=True, return_dict=True)
    return ids, second, margin, top1, logits0


def case_decode_7b():
    """Greedy decode at the shape BASELINE configs[2] names: the full-depth model (ViT-H x 32 + 7B-shaped InternLM2 x 32, the weights of
    `full_depth`), the four tiles of `full_depth` as a batch of four prompts of S = 1081 (one left-padded), 16 new tokens through
    InternVLSAMModel.generate's embedding path (modeling_internvl_sam.py:394-431) and the reference's LLM forward with its KV cache.
    Stores ids, the runner-up id and the top-2 logit margin of every step (random 7B-shaped weights give near-ties: a bf16 run can only be
    held to the reference's token where the margin allows)."""
    from ullsam_amd.utils.synthetic import microscopy_batch
    t = time.time()
    m = _build_full_depth()
    fill_module_inplace(m, seed=0)
    print(f"  built + filled in {time.time() - t:.0f}s", flush=True)
    ids, mask = decode_7b_ids()
    tids, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
    x_np, _ = microscopy_batch(FULL_TILE_SEEDS)
    t = time.time()
    vit_embeds = torch.cat([m.extract_feature(torch.from_numpy(x_np[b:b + 1]))[0] for b in range(4)])    # [4, 1024, 4096]; per tile: memory
    print(f"  extract_feature x 4 in {time.time() - t:.0f}s", flush=True)
    emb = m.language_model.get_input_embeddings()(tids).clone()
    B, N, C = emb.shape
    emb = emb.reshape(B * N, C)
    sel = tids.reshape(-1) == 92546
    emb[sel] = vit_embeds.reshape(-1, C)[: int(sel.sum())]                                                # :407-421
    emb = emb.reshape(B, N, C)
    t = time.time()
    out_ids, second, margin, top1, logits0 = _ref_greedy_batched(m.language_model, emb, tmask, DECODE_7B_NEW)
    print(f"  greedy x {DECODE_7B_NEW} in {time.time() - t:.0f}s")
    print("  ids\n", out_ids, "\n  margins\n", margin.round(4), flush=True)
    save("decode_7b", weight_seed=0, tile_seeds=np.asarray(FULL_TILE_SEEDS, np.int64), ids=ids, mask=mask, greedy_ids=out_ids, second_ids=second,
         margin=margin, top1=top1, logits0_sample=logits0[:, ::97].copy(), vit_embeds_sample=vit_embeds.numpy().reshape(-1)[::FULL_STRIDE].copy())


DECODE_7B_FORCED_SEED = 77


def decode_7b_forced_ids():
    """-> [4, 16] int64: the teacher-forced continuation of tests/golden/decode_7b_forced.npz -- seeded random ids from the body of the
    vocabulary (no eos / pad / image-context id), DIFFERENT at every step and prompt (the greedy fixture's random-weight rows repeat one id)."""
    return np.random.default_rng(DECODE_7B_FORCED_SEED).integers(1000, 90000, size=(4, DECODE_7B_NEW)).astype(np.int64)


def _ref_forced(lm, emb, mask, forced):
    """Prefill + one cached forward per forced id (modeling_internlm2.py:1112-1149, 383-426) -> (prefill output, [B, n, V] fp32 logits AFTER each forced id)."""
    B = emb.shape[0]
    pos = mask.long().cumsum(-1) - 1
    pos.masked_fill_(mask == 0, 1)
    o0 = lm(inputs_embeds=emb, attention_mask=mask, position_ids=pos, use_cache=True, return_dict=True)
    o, rows = o0, []
    for s in range(forced.shape[1]):
        mask = torch.cat([mask, torch.ones((B, 1), dtype=mask.dtype)], 1)
        p1 = (mask.long().cumsum(-1) - 1)[:, -1:]
        o = lm(input_ids=torch.from_numpy(forced[:, s:s + 1]).contiguous(), attention_mask=mask, position_ids=p1, past_key_values=o.past_key_values, use_cache=True, return_dict=True)
        rows.append(o.logits[:, -1].float().numpy().copy())
    return o0, np.stack(rows, 1)


def case_decode_7b_forced():
    """What decode_7b cannot pin (its greedy rows are low-entropy and carry logits for step 0 only): the same model, prompts and tiles, but
      (a) TEACHER-FORCED: 16 seeded random continuation ids per prompt fed through the reference's cached forward -- per step a strided
          sample of the fp32 logits, the top-1 / runner-up ids and their margin (64 decode steps whose numerics are pinned directly:
          cache append, attention over a growing cache, the skinny GEMMs at the 7B shape);
      (b) the same passes under torch.autocast("cpu", bfloat16) (ViT included): per-step mean / max |logits - fp32 logits| over the whole
          vocabulary -- the bound the bf16 mode's decode steps are held to -- and the reference's OWN autocast greedy ids for these prompts."""
    from ullsam_amd.utils.synthetic import microscopy_batch
    t = time.time()
    m = _build_full_depth()
    fill_module_inplace(m, seed=0)
    print(f"  built + filled in {time.time() - t:.0f}s", flush=True)
    ids, mask = decode_7b_ids()
    forced = decode_7b_forced_ids()
    tids, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
    x_np, _ = microscopy_batch(FULL_TILE_SEEDS)
    sel = tids.reshape(-1) == 92546

    def embed():
        vit = torch.cat([m.extract_feature(torch.from_numpy(x_np[b:b + 1]))[0] for b in range(4)]).float()
        emb = m.language_model.get_input_embeddings()(tids).clone().float()
        B, N, C = emb.shape
        emb = emb.reshape(B * N, C)
        emb[sel] = vit.reshape(-1, C)[: int(sel.sum())]
        return emb.reshape(B, N, C)

    t = time.time()
    emb = embed()
    _, lg = _ref_forced(m.language_model, emb, tmask, forced)
    print(f"  fp32: embeds + prefill + {DECODE_7B_NEW} forced steps in {time.time() - t:.0f}s", flush=True)
    v, i = torch.from_numpy(lg).topk(2, -1)
    t = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16, cache_enabled=False):
        emb_ac = embed()
        o0, lg_ac = _ref_forced(m.language_model, emb_ac, tmask, forced)
        print(f"  autocast: embeds + prefill + forced steps in {time.time() - t:.0f}s", flush=True)
        t = time.time()
        ac_ids, _, ac_margin, _, _ = _ref_greedy_batched(m.language_model, emb_ac, tmask, DECODE_7B_NEW)
        print(f"  autocast greedy in {time.time() - t:.0f}s", flush=True)
    d = np.abs(lg_ac.astype(np.float64) - lg)
    print("  forced top-1 ids\n", i[..., 0].numpy(), "\n  margins\n", (v[..., 0] - v[..., 1]).numpy().round(4))
    print("  autocast mean |dlogit| per step\n", d.mean(-1).round(4), "\n  autocast greedy ids\n", ac_ids, flush=True)
    save("decode_7b_forced", weight_seed=0, tile_seeds=np.asarray(FULL_TILE_SEEDS, np.int64), ids=ids, mask=mask, forced_ids=forced,
         logits_sample=lg[:, :, ::97].copy(), logits_absmax=np.abs(lg).max(-1), top1_ids=i[..., 0].numpy(), second_ids=i[..., 1].numpy(),
         margin=(v[..., 0] - v[..., 1]).numpy(), top1=v[..., 0].numpy(), ac_mean_err=d.mean(-1), ac_max_err=d.max(-1),
         ac_top1_ids=lg_ac.argmax(-1), ac_greedy_ids=ac_ids, ac_greedy_margin=ac_margin)


SAM_H_SEEDS = (3, 5)


def case_sam_h_forward():
    """BASELINE configs[1] through the reference: sam_model_registry['vit_h']() (build_sam.py:14-21: ViT-H x 32 + prompt encoder + mask
    decoder), Sam.forward (sam.py:53-131) on two 1024^2 microscopy tiles given as 0..255 images with one positive click each, single-mask
    output; fp32 and under torch.autocast(bfloat16) (the bound of the bf16 mode)."""
    from build_sam import sam_model_registry
    from ullsam_amd.utils.synthetic import microscopy_batch
    t = time.time()
    saved = (torch.nn.Linear.reset_parameters, torch.nn.init.trunc_normal_)
    torch.nn.Linear.reset_parameters = lambda self: None
    torch.nn.init.trunc_normal_ = lambda t_, *a, **k: t_
    try:
        sam = sam_model_registry["vit_h"]()
    finally:
        torch.nn.Linear.reset_parameters, torch.nn.init.trunc_normal_ = saved
    fill_module_inplace(sam, seed=0)
    print(f"  built + filled in {time.time() - t:.0f}s", flush=True)
    x_np, pts = microscopy_batch(SAM_H_SEEDS)
    lbl = np.ones((1, 1), np.int32)

    def run():
        recs = [{"image": torch.from_numpy(x_np[b] * 255.0), "original_size": (1024, 1024), "point_coords": torch.from_numpy(pts[b:b + 1]),
                 "point_labels": torch.from_numpy(lbl)} for b in range(len(SAM_H_SEEDS))]
        emb = {}
        h = sam.image_encoder.register_forward_hook(lambda mod, inp, out: emb.__setitem__("e", out.detach().float().numpy().copy()))
        res = sam(recs, multimask_output=False)
        h.remove()
        return emb["e"], res

    t = time.time()
    e32, r32 = run()
    t32 = time.time() - t
    with torch.autocast("cpu", dtype=torch.bfloat16, cache_enabled=False):
        e16, r16 = run()
    print(f"  fp32 {t32:.0f}s, autocast {time.time() - t - t32:.0f}s", flush=True)
    out = {}
    for b in range(len(SAM_H_SEEDS)):
        low, low16 = r32[b]["low_res_logits"].float().numpy(), r16[b]["low_res_logits"].float().numpy()
        mk, mk16 = r32[b]["masks"].numpy(), r16[b]["masks"].numpy()
        out[f"low_{b}"] = low
        out[f"iou_pred_{b}"] = r32[b]["iou_predictions"].float().numpy()
        out[f"mask_bits_{b}"] = np.packbits(mk)
        out[f"mask_fill_{b}"] = np.float64(mk.mean())
        out[f"img_emb_{b}"] = e32[b].reshape(-1)[::37].copy()
        out[f"img_emb_{b}_ac_mean_err"] = np.float64(np.abs(e16[b] - e32[b]).mean())
        out[f"low_{b}_ac_mean_err"] = np.float64(np.abs(low16 - low).mean())
        out[f"low_{b}_ac_max_err"] = np.float64(np.abs(low16 - low).max())
        out[f"ac_mask_iou_{b}"] = np.float64(O.calc_iou(mk16, mk))
        print(f"  tile {SAM_H_SEEDS[b]}: mask fill {mk.mean():.4f}, logits mean|x| {np.abs(low).mean():.3f}, autocast: logits mean err {out[f'low_{b}_ac_mean_err']:.5f} "
              f"max {out[f'low_{b}_ac_max_err']:.4f}, mask IoU {out[f'ac_mask_iou_{b}']:.6f}", flush=True)
    save("sam_h_forward", weight_seed=0, tile_seeds=np.asarray(SAM_H_SEEDS, np.int64), pts=pts, lbl=lbl, **out)


def case_train_slice(boxes: bool = False):
    """(boxes=True: every instance also carries a BOX prompt, as the trainer forwards `boxes=boxes`, train_joint_v2.py:975,1038,1057 -> no pad point,
    two corner embeddings with point_embeddings[2] / [3], prompt_encoder.py:96-103,181 -> fixture train_slice_box.)
    Gradients of the reference's segmentation loss (train_joint_v2.py:1026-1100: text_aware_dense_feature -> prompt encoder -> mask
    decoder -> bilinear upsample -> BCE + Dice, calc_instance_loss :774-812) with respect to every parameter downstream of the LLM's last
    hidden state -- mlp2, the prompt encoder, the mask decoder -- on the `ullsam_tiny` composite (full-size decoder: 64 x 64 image tokens,
    256 channels), two instances with two clicks each.  The LLM hidden states and the image embedding are inputs (seeded), as they are
    constants for this slice.  train_joint_v2 imports torchvision / PIL / wandb at module level, which this container lacks and the
    loss code does not use: empty stand-in modules are registered for the import only."""
    import sys, types
    from transformers import AutoTokenizer, GenerationConfig, get_cosine_schedule_with_warmup, AutoModel, AutoConfig  # noqa: F401 (resolved before the stand-ins exist: transformers probes for torchvision lazily)
    for name in ("torchvision", "torchvision.transforms", "wandb", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    import train_joint_v2 as TJ
    from modeling.configuration_internvl_chat import InternVLChatConfig
    from modeling.modeling_internvl_sam import InternVLSAMModel
    sam = _sam_small()
    cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT-B-16"]}, llm_config=dict(LLM_TINY),
                             downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
    cfg.llm_config.rope_scaling = None
    m = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)
    fill_module(m, seed=0)
    m.train()
    for p_ in m.parameters():
        p_.requires_grad_(True)
    rng = np.random.default_rng(11)
    hid = rng.standard_normal((1, 1024, LLM_TINY["hidden_size"]), dtype=np.float32)
    img = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    pts = np.array([[[300.0, 340.0], [120.0, 800.0]], [[700.0, 610.0], [64.0, 64.0]]], np.float32)     # [instances, clicks, 2]
    lbl = np.array([[1, 0], [1, 1]], np.int32)
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    with torch.enable_grad():
        last = m.text_aware_dense_feature(torch.from_numpy(hid))                      # [1, 256, 64, 64]
        bs = pts.shape[0]
        last = last.repeat(bs, 1, 1, 1)                                               # train_joint_v2.py:1052-1054
        bxs = np.array([[150.0, 190.0, 450.0, 490.0], [480.0, 390.0, 920.0, 830.0]], np.float32) if boxes else None   # x0, y0, x1, y1 around the two discs
        sp, de = m.prompt_encoder(points=(torch.from_numpy(pts), torch.from_numpy(lbl)), boxes=None if bxs is None else torch.from_numpy(bxs), masks=None, llm_hidden_states=last)
        low, iou = m.mask_decoder(image_embeddings=torch.from_numpy(img), image_pe=m.prompt_encoder.get_dense_pe(),
                                  sparse_prompt_embeddings=sp, dense_prompt_embeddings=de, multimask_output=False)
        pred = torch.nn.functional.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)
        loss, bce, dice, iou_val = TJ.calc_instance_loss(pred, torch.from_numpy(gt), TJ.BCELoss(), TJ.DiceLoss())
        loss.backward()
    out = {"hid_sample": hid.reshape(-1)[::1009].copy(), "img_seed": 11, "pts": pts, "lbl": lbl, "loss": np.float32(loss.item()), "bce": np.float32(bce.item()),
           "dice": np.float32(dice.item()), "low_sample": low.detach().numpy().reshape(-1)[::61].copy()}
    if boxes:
        out["boxes"] = bxs
    names = []
    for name, p_ in m.named_parameters():
        if not name.startswith(("mlp2.", "prompt_encoder.", "mask_decoder.")) or p_.grad is None:
            continue
        g = p_.grad.numpy().reshape(-1)
        stride = max(1, g.size // 2048)
        names.append(name)
        out["g:" + name] = g[::stride].copy()
        out["n:" + name] = np.float32(np.sqrt((g.astype(np.float64) ** 2).sum()))
    out["names"] = np.array(names)
    save("train_slice_box" if boxes else "train_slice", **out)


def case_train_llm_slice(pad: int = 0):
    """(pad > 0: the prompt is LEFT-PADDED by `pad` positions with attention_mask = 0 there -- the additive finfo.min padding mask of
    modeling_internlm2.py:114-125 in the training attention's forward and backward -> fixture train_llm_slice_pad.)
    As case_train_slice, but the LLM hidden states are no longer an input: the vision features (seeded, constant -- the reference
    computes them under no_grad, modeling_internvl_sam.py:243-244) go through pixel_shuffle + mlp1, are spliced into the token embeddings
    (:136-158), run through the tiny InternLM2 (2 layers, frozen) and come out as hidden_states[-1] over the image tokens (:195-205) before
    the segmentation branch.  Stored: the loss, the gradients of mlp1 (reached only through the LLM's backward), of mlp2 and of a few
    decoder tensors, and a sample of d loss / d vision features."""
    import sys, types
    from transformers import AutoTokenizer, GenerationConfig, get_cosine_schedule_with_warmup, AutoModel, AutoConfig  # noqa: F401
    for name in ("torchvision", "torchvision.transforms", "wandb", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    import train_joint_v2 as TJ
    from modeling.configuration_internvl_chat import InternVLChatConfig
    from modeling.modeling_internvl_sam import InternVLSAMModel
    sam = _sam_small()
    cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT-B-16"]}, llm_config=dict(LLM_TINY),
                             downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
    cfg.llm_config.rope_scaling = None
    m = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)
    fill_module(m, seed=0)
    m.train()
    for n_, p_ in m.named_parameters():
        p_.requires_grad_(not n_.startswith(("language_model.", "vision_model.")))
    rng = np.random.default_rng(12)
    feat = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)              # vision_model output (NCHW)
    img = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)               # image embedding for the decoder (second ViT call in the trainer)
    ids = O.make_input_ids(n_text_pre=20, n_text_post=34, seed=1)
    amask = np.ones_like(ids)
    if pad:
        ids = np.concatenate([np.full((1, pad), 2, ids.dtype), ids], 1)          # pad token id 2, masked out
        amask = np.concatenate([np.zeros((1, pad), amask.dtype), amask], 1)
    tids = torch.from_numpy(ids)
    pts = np.array([[[300.0, 340.0], [120.0, 800.0]], [[700.0, 610.0], [64.0, 64.0]]], np.float32)
    lbl = np.array([[1, 0], [1, 1]], np.int32)
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    with torch.enable_grad():
        vf = torch.from_numpy(feat).requires_grad_(True)
        features = m.pixel_shuffle(vf.permute(0, 2, 3, 1), scale_factor=m.downsample_ratio)
        features = features.reshape(features.shape[0], -1, features.shape[-1])
        vit_embeds = m.mlp1(features)                                            # extract_feature :245-249
        emb = m.language_model.get_input_embeddings()(tids).clone()
        B, N, C = emb.shape
        emb = emb.reshape(B * N, C)
        selected = tids.reshape(-1) == m.img_context_token_id
        emb[selected] = emb[selected] * 0.0 + vit_embeds.reshape(-1, C)          # forward :136-152
        emb = emb.reshape(B, N, C)
        outputs = m.language_model(inputs_embeds=emb, attention_mask=torch.from_numpy(amask), output_hidden_states=True, return_dict=True)
        tok_idx = torch.nonzero(selected.reshape(B, N), as_tuple=True)[1]
        hidden = outputs.hidden_states[-1][:, int(tok_idx.min()):int(tok_idx.max()) + 1, :]
        last = m.text_aware_dense_feature(hidden)
        bs = pts.shape[0]
        last = last.repeat(bs, 1, 1, 1)
        sp, de = m.prompt_encoder(points=(torch.from_numpy(pts), torch.from_numpy(lbl)), boxes=None, masks=None, llm_hidden_states=last)
        low, iou = m.mask_decoder(image_embeddings=torch.from_numpy(img), image_pe=m.prompt_encoder.get_dense_pe(),
                                  sparse_prompt_embeddings=sp, dense_prompt_embeddings=de, multimask_output=False)
        pred = torch.nn.functional.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)
        loss, bce, dice, iou_val = TJ.calc_instance_loss(pred, torch.from_numpy(gt), TJ.BCELoss(), TJ.DiceLoss())
        loss.backward()
    out = {"seed": 12, "ids": ids, "attention_mask": amask, "pts": pts, "lbl": lbl, "loss": np.float32(loss.item()), "bce": np.float32(bce.item()),
           "dice": np.float32(dice.item()), "hidden_sample": hidden.detach().numpy().reshape(-1)[::97].copy(),
           "g:vit_features": vf.grad.numpy().reshape(-1)[::257].copy(),
           "n:vit_features": np.float32(np.sqrt((vf.grad.numpy().astype(np.float64) ** 2).sum()))}
    names = []
    keep = ("mlp1.", "mlp2.", "prompt_encoder.llm", "mask_decoder.transformer.layers.0.self_attn.q_proj", "mask_decoder.output_upscaling.0")
    for name, p_ in m.named_parameters():
        if not name.startswith(keep) or p_.grad is None:
            continue
        g = p_.grad.numpy().reshape(-1)
        stride = max(1, g.size // 2048)
        names.append(name)
        out["g:" + name] = g[::stride].copy()
        out["n:" + name] = np.float32(np.sqrt((g.astype(np.float64) ** 2).sum()))
    out["names"] = np.array(names)
    save("train_llm_slice_pad" if pad else "train_llm_slice", **out)


def case_train_vit_slice():
    """As case_train_slice, with the image embedding no longer an input: a seeded image goes through the reference's vision model
    (`_sam_small`: 2 blocks of width 128, one 14x14-windowed with 25 padded windows, one global over 64 x 64 tokens; neck) with gradients,
    as the trainer's second ViT call does (train_joint_v2.py:1014-1021).  Stored: the loss and the gradient of every vision-model parameter."""
    import sys, types
    from transformers import AutoTokenizer, GenerationConfig, get_cosine_schedule_with_warmup, AutoModel, AutoConfig  # noqa: F401
    for name in ("torchvision", "torchvision.transforms", "wandb", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    import train_joint_v2 as TJ
    from modeling.configuration_internvl_chat import InternVLChatConfig
    from modeling.modeling_internvl_sam import InternVLSAMModel
    sam = _sam_small()
    cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT-B-16"]}, llm_config=dict(LLM_TINY),
                             downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
    cfg.llm_config.rope_scaling = None
    m = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)
    fill_module(m, seed=0)
    m.train()
    for n_, p_ in m.named_parameters():
        p_.requires_grad_(not n_.startswith("language_model."))
    rng = np.random.default_rng(13)
    hid = rng.standard_normal((1, 1024, LLM_TINY["hidden_size"]), dtype=np.float32)
    x = rand_image((1, 3, 1024, 1024), seed=13)
    pts = np.array([[[300.0, 340.0], [120.0, 800.0]], [[700.0, 610.0], [64.0, 64.0]]], np.float32)
    lbl = np.array([[1, 0], [1, 1]], np.int32)
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    with torch.enable_grad():
        image_embeddings = m.vision_model(torch.from_numpy(x))
        last = m.text_aware_dense_feature(torch.from_numpy(hid)).repeat(pts.shape[0], 1, 1, 1)
        sp, de = m.prompt_encoder(points=(torch.from_numpy(pts), torch.from_numpy(lbl)), boxes=None, masks=None, llm_hidden_states=last)
        low, iou = m.mask_decoder(image_embeddings=image_embeddings, image_pe=m.prompt_encoder.get_dense_pe(),
                                  sparse_prompt_embeddings=sp, dense_prompt_embeddings=de, multimask_output=False)
        pred = torch.nn.functional.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)
        loss, bce, dice, iou_val = TJ.calc_instance_loss(pred, torch.from_numpy(gt), TJ.BCELoss(), TJ.DiceLoss())
        loss.backward()
    out = {"seed": 13, "hid_sample": hid.reshape(-1)[::1009].copy(), "pts": pts, "lbl": lbl, "loss": np.float32(loss.item()), "bce": np.float32(bce.item()),
           "dice": np.float32(dice.item()), "emb_sample": image_embeddings.detach().numpy().reshape(-1)[::997].copy()}
    names = []
    for name, p_ in m.named_parameters():
        if not name.startswith("vision_model.") or p_.grad is None:
            continue
        g = p_.grad.numpy().reshape(-1)
        stride = max(1, g.size // 2048)
        names.append(name)
        out["g:" + name] = g[::stride].copy()
        out["n:" + name] = np.float32(np.sqrt((g.astype(np.float64) ** 2).sum()))
    out["names"] = np.array(names)
    save("train_vit_slice", **out)


def case_train_step(real_dims: bool = False):
    """(real_dims: the same step on a composite with the bench configuration's HEAD DIMENSIONS -- SAM ViT-B width (768 = 12 heads x 64, one
    windowed + one global block on the 64 x 64 token grid) and ONE 7B-shaped InternLM2 layer (hidden 4096, 32 heads / 8 KV heads x 128,
    intermediate 14336) -> fixture train_step_real.)
    One whole step of the reference's trainer on the `ullsam_tiny` composite, as train_joint_v2.py:990-1100 runs it: model(pixel_values,
    input_ids, ..., output_hidden_states=True) -> outputs.hidden_states (the text-aware dense feature), image_embeddings =
    model.vision_model(pixel_values) with gradients, prompt encoder, mask decoder, upsample, calc_instance_loss; LLM frozen, everything else
    trainable (setup_model_params :1280-1359).  Stored: the loss and a sample + norm of the gradient of EVERY trainable parameter."""
    import sys, types
    from transformers import AutoTokenizer, GenerationConfig, get_cosine_schedule_with_warmup, AutoModel, AutoConfig  # noqa: F401
    for name in ("torchvision", "torchvision.transforms", "wandb", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    import train_joint_v2 as TJ
    from modeling.configuration_internvl_chat import InternVLChatConfig
    from modeling.modeling_internvl_sam import InternVLSAMModel
    sam = _sam_small(depth=2, embed_dim=768, heads=12, glob=(1,)) if real_dims else _sam_small()
    cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT-B-16"]}, llm_config=dict(LLM_7B_L1) if real_dims else dict(LLM_TINY),
                             downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
    cfg.llm_config.rope_scaling = None
    m = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)
    if real_dims:
        fill_module_inplace(m, seed=0)
    else:
        fill_module(m, seed=0)
    m.train()
    for n_, p_ in m.named_parameters():
        p_.requires_grad_(not n_.startswith("language_model."))
    x = torch.from_numpy(rand_image((1, 3, 1024, 1024), seed=14))
    ids = O.make_input_ids(n_text_pre=20, n_text_post=34, seed=1)
    tids = torch.from_numpy(ids)
    pts = np.array([[[300.0, 340.0], [120.0, 800.0]], [[700.0, 610.0], [64.0, 64.0]]], np.float32)
    lbl = np.array([[1, 0], [1, 1]], np.int32)
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    with torch.enable_grad():
        outputs = m(pixel_values=x, input_ids=tids, attention_mask=torch.ones_like(tids), image_flags=(tids == 92546)[..., None].long(),
                    return_dict=True, use_cache=False, output_hidden_states=True)
        last = outputs.hidden_states                                              # train_joint_v2.py:1010
        image_embeddings = m.vision_model(x)                                      # :1020
        bs = pts.shape[0]
        last = last.repeat(bs, 1, 1, 1)
        sp, de = m.prompt_encoder(points=(torch.from_numpy(pts), torch.from_numpy(lbl)), boxes=None, masks=None, llm_hidden_states=last)
        low, iou = m.mask_decoder(image_embeddings=image_embeddings, image_pe=m.prompt_encoder.get_dense_pe(),
                                  sparse_prompt_embeddings=sp, dense_prompt_embeddings=de, multimask_output=False)
        pred = torch.nn.functional.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)
        loss, bce, dice, iou_val = TJ.calc_instance_loss(pred, torch.from_numpy(gt), TJ.BCELoss(), TJ.DiceLoss())
        loss.backward()
    out = {"seed": 14, "ids": ids, "pts": pts, "lbl": lbl, "loss": np.float32(loss.item()), "bce": np.float32(bce.item()), "dice": np.float32(dice.item())}
    names = []
    for name, p_ in m.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad.numpy().reshape(-1)
        stride = max(1, g.size // 512)
        names.append(name)
        out["g:" + name] = g[::stride].copy()
        out["n:" + name] = np.float32(np.sqrt((g.astype(np.float64) ** 2).sum()))
    out["names"] = np.array(names)
    save("train_step_real" if real_dims else "train_step", **out)


def mask_prompt_inputs():
    """-> (masks [2, 1, 256, 256] fp32, weights R [2, 256, 64, 64] fp32 of the scalar the gradients are taken of, points, labels): seed-derived, restated by tests/test_train_gpu.py."""
    rng = np.random.default_rng(21)
    masks = (rng.standard_normal((2, 1, 256, 256), dtype=np.float32) * 2.0 + 0.25).astype(np.float32)
    R = rng.standard_normal((2, 256, 64, 64), dtype=np.float32)
    pts = np.array([[[300.0, 340.0]], [[700.0, 610.0]]], np.float32)
    lbl = np.array([[1], [0]], np.int32)
    return masks, R, pts, lbl


def case_train_mask_prompt():
    """Gradients THROUGH A MASK PROMPT (prompt_encoder.py:54-62 mask_downscaling, :105-108 _embed_masks, :187-188): the reference's PromptEncoder with a point and a mask
    per prompt, the scalar sum(dense_embeddings * R) / numel for a seeded R; its autograd gradients with respect to every parameter of mask_downscaling and to the masks."""
    sam = _sam_small()
    pe = sam.prompt_encoder
    fill_module(pe, seed=0, prefix="prompt_encoder.")
    pe.train()
    for p_ in pe.parameters():
        p_.requires_grad_(True)
    masks_np, R, pts, lbl = mask_prompt_inputs()
    masks = torch.from_numpy(masks_np).requires_grad_(True)
    with torch.enable_grad():
        sparse, dense = pe(points=(torch.from_numpy(pts), torch.from_numpy(lbl)), boxes=None, masks=masks, llm_hidden_states=None)
        loss = (dense * torch.from_numpy(R)).sum() / dense.numel()
        loss.backward()
    out = {"loss": np.float64(loss.item()), "dense_sample": dense.detach().numpy().reshape(-1)[::997].copy(), "sparse": sparse.detach().numpy().copy(),
           "g_masks_sample": masks.grad.numpy().reshape(-1)[::61].copy(), "g_masks_absmax": np.float64(masks.grad.abs().max().item())}
    names = []
    for name, p_ in pe.named_parameters():
        if name.startswith("mask_downscaling"):
            assert p_.grad is not None, name
            names.append("prompt_encoder." + name)
            out["g:prompt_encoder." + name] = p_.grad.numpy().reshape(-1).copy()
    out["names"] = np.array(names)
    print("  loss", loss.item(), "; gradients of", len(names), "tensors; |d masks| max", float(masks.grad.abs().max()))
    save("train_mask_prompt", **out)


CASES = {"train_step": case_train_step, "train_step_real": lambda: case_train_step(real_dims=True), "train_vit_slice": case_train_vit_slice, "train_slice": case_train_slice, "train_slice_box": lambda: case_train_slice(boxes=True), "train_mask_prompt": case_train_mask_prompt, "train_llm_slice": case_train_llm_slice, "train_llm_slice_pad": lambda: case_train_llm_slice(pad=37), "chat_prompt": case_chat_prompt, "amg": case_amg, "vit_tiny": case_vit_tiny, "vit_tiny_relpos_interp": case_vit_tiny_relpos_interp, "decoder": case_decoder, "llm_tiny": case_llm_tiny,
         "ullsam_tiny": case_ullsam_tiny, "sam_forward": case_sam_forward, "vit_b_full": case_vit_b_full,
         "vit_h_d2": case_vit_h_d2, "llm_7b_l1": case_llm_7b_l1,
         "rope_variants": case_rope_variants, "llm_tiny_bias_linear": case_llm_tiny_bias_linear,
         "full_depth": case_full_depth, "decode_7b": case_decode_7b, "decode_7b_forced": case_decode_7b_forced, "sam_h_forward": case_sam_h_forward}

if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES)):
        print(f"[gen_golden] {n}")
        t0 = time.time()
        CASES[n]()
        print(f"  done in {time.time() - t0:.1f}s")
