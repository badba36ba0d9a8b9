"""Headline benchmark: images/s end-to-end through the uLLSAM mask path (app.py:580-645 call sequence) on MI355X.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`, WORLD_SIZE set), or -- when WORLD_SIZE is unset -- this file starts that launcher itself as a child
process BEFORE anything touches the GPU and relays its JSON line.

A "step" is one pass of the hot path over one batch of synthetic 1024x1024 tiles resident in HBM:
  InternVLSAMModel.forward (SAM ViT encoder -> pixel-shuffle + mlp1 -> InternLM2 prefill -> mlp2 + inverse shuffle)
  -> PromptEncoder.forward (1 point + LLM dense prompt) -> MaskDecoder.forward -> x4 bilinear upsample -> threshold
  -> (N > 1) RCCL all-gather of low-res logits + masks.
Default workload = BASELINE.json configs[2] per GPU (ViT-H + InternLM2-7B-shaped + mask decoder, batch 4, bf16, S = 1081);
N GPUs process N x 4 images (weak scaling; configs[3] at N = 8).  Weights are random-init (no checkpoints exist offline).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the MFMA GEMM; achieved = algorithmic FLOPs of every GEMM
launch / their HIP-event durations, sampled on every 5th timed step on the launch stream: an event pair per GEMM on EVERY step cost 1.5 ms
of a 78 ms step -- rounds 1 - 4 carried that), `step_ms` (min / median / max of the timed steps' own event spans: box and run variance in the record),
`graph_ms_per_step` (the step replayed from ONE HIP-graph capture, untimed: eager minus this = the cost of launching kernel by kernel)
and `cpu_baseline` (the oracle restated on torch CPU tensors -- multi-threaded fp32 GEMMs, `kind:
"port-torch"` -- timed on the host cores on a bounded, depth-reduced sample of the same workload, the numpy oracle's figure beside it; N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIT = {"h": dict(dim=1280, depth=32, heads=16, glob=[7, 15, 23, 31]), "l": dict(dim=1024, depth=24, heads=16, glob=[5, 11, 17, 23]),
       "b": dict(dim=768, depth=12, heads=12, glob=[2, 5, 8, 11])}
LLM = {"7b": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8),
       "2b": dict(hidden_size=2048, intermediate_size=8192, num_hidden_layers=24, num_attention_heads=16, num_key_value_heads=8),
       "none": None}
DATA_NOTE = "synthetic (1024x1024 microscopy tiles: flat-intensity cells + noise; random-init weights = the values of the reference-generated parity fixtures; synthetic token ids)"
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)


FIXTURE_TILE_SEEDS = (3, 5, 27, 32)   # the tiles of tests/golden/full_depth.npz (fp32 mask fill 0.95 / 0.53 / 0.48 / 0.60 in the reference)


def init_random_(model: torch.nn.Module, seed: int = 0):
    """Random-init weights (no checkpoints exist offline): the VALUES the reference-generated parity fixtures were filled with
    (ullsam_amd.utils.synthetic.fixture_param == oracle.fill_param, bit for bit: tests/test_host_cpu.py), generated on the host's cores and
    copied to the device.  With seed 0 the model IS the model of tests/golden/full_depth.npz, so the masks of rank 0's first four tiles
    can be scored against the reference's own fp32 masks stored there (`mask_iou_vs_reference`)."""
    from ullsam_amd.utils.synthetic import fill_model_like_fixtures
    with torch.no_grad():
        fill_model_like_fixtures(model, seed, workers=min(64, os.cpu_count() or 8))


def build_model(vit: str, llm: str, dtype: torch.dtype, device: str, init: bool = True):
    from ullsam_amd.build_sam import _build_sam
    from ullsam_amd.modeling.configuration_internvl_chat import InternVLChatConfig
    from ullsam_amd.modeling.modeling_internvl_sam import InternVLSAMModel
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            v = VIT[vit]
            sam = _build_sam(v["dim"], v["depth"], v["heads"], v["glob"])
            if LLM[llm] is None:
                model = sam
            else:
                cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT"]},
                                         llm_config=dict(architectures=["InternLM2ForCausalLM"], vocab_size=92553, bias=False,
                                                         max_position_embeddings=32768, rope_theta=1000000, rms_norm_eps=1e-5, **LLM[llm]),
                                         downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
                model = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)
    finally:
        torch.set_default_dtype(old)
    model = model.to(device).to(dtype).eval()
    if init:
        init_random_(model)
    return model


def make_input_ids(n_text_pre: int, n_text_post: int, n_img: int = 1024, seed: int = 1, batch: int = 1) -> np.ndarray:
    """Synthetic prompt ids with the layout chat() builds (modeling_internvl_sam.py:296-312; the tokenizer cannot be loaded
    offline): bos, text, <img> = 92544, n_img x <IMG_CONTEXT> = 92546, </img> = 92545, text."""
    rng = np.random.default_rng(seed)
    rows = [np.concatenate([[1], rng.integers(3, 92000, n_text_pre), [92544], np.full(n_img, 92546), [92545],
                            rng.integers(3, 92000, n_text_post)]) for _ in range(batch)]
    return np.asarray(rows, np.int64)


GEMM_EVENT_EVERY = 5   # timed steps between two steps whose GEMM launches carry event pairs (see main)


class GemmTimer:
    """HIP-event pairs around every ullsam_gemm launch on the launch stream (torch's current stream)."""

    def __init__(self):
        from ullsam_amd import ops
        self.ops, self.orig, self.rec, self.on = ops, ops.gemm, [], False
        self.alg_bytes = 0  # operand + result bytes of every timed launch (each read / written once)

        def timed(a, w, *args, **kw):
            if not self.on:
                return self.orig(a, w, *args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.orig(a, w, *args, **kw)
            e1.record()
            self.rec.append((e0, e1, 2.0 * a.shape[0] * w.shape[0] * a.shape[1]))
            self.alg_bytes += a.numel() * a.element_size() + w.numel() * w.element_size() + out.numel() * out.element_size()
            return out

        ops.gemm = timed
        orig_rope = ops.gemm_qkv_rope   # the wqkv GEMM with the RoPE epilogue is a GEMM launch like the others

        def timed_rope(x, wqkv, *args, **kw):
            if not self.on:
                return orig_rope(x, wqkv, *args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_rope(x, wqkv, *args, **kw)
            e1.record()
            self.rec.append((e0, e1, 2.0 * x.shape[0] * wqkv.shape[0] * x.shape[1]))
            self.alg_bytes += x.numel() * x.element_size() + wqkv.numel() * wqkv.element_size() + x.shape[0] * wqkv.shape[0] * x.element_size()
            return out

        ops.gemm_qkv_rope = timed_rope

    def summary(self):
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.rec)
        fl = sum(f for _, _, f in self.rec)
        return len(self.rec), ms, fl


def tile_seeds(rank: int, B: int):
    """Rank 0's first four tiles are the tiles of the full-depth parity fixture; every other image has its own seed."""
    return [FIXTURE_TILE_SEEDS[b] if (rank == 0 and b < len(FIXTURE_TILE_SEEDS)) else 1000 * rank + 100 + b for b in range(B)]


def make_inputs(B, S, device, full: bool):
    """Synthetic microscopy tiles (flat-intensity cells on a dark background + noise: ullsam_amd/utils/synthetic.py) resident in HBM, one
    positive click per image at a cell centre, synthetic prompt ids.  Different tiles on every rank."""
    from ullsam_amd.utils.synthetic import microscopy_batch
    rank = int(os.environ.get("RANK", "0"))
    x_np, pts_np = microscopy_batch(tile_seeds(rank, B))
    x = torch.from_numpy(x_np).to(device)
    pts = torch.from_numpy(pts_np).to(device)
    lbl = torch.ones((B, 1), dtype=torch.int32, device=device)
    ids = None
    if full:
        # ONE prompt for every image (app.py asks the same question of every tile; it is also the prompt of tests/golden/full_depth.npz)
        ids = torch.from_numpy(np.repeat(make_input_ids(20, S - 1047, seed=1, batch=1), B, 0)).to(device)
        assert ids.shape == (B, S)
    return x, pts, lbl, ids


def mask_path_compute(model, inputs, dtype):
    """One pass of the hot path over the batch (app.py:580-645's call sequence; prompt encoder / mask decoder / upsample run once
    over the B images -- one image per prompt) -> (low-res logits, thresholded masks).  One HIP stream: cutting the batch into shards on
    2 / 4 streams was measured again in round 4 (tools/probes/streams_ab.py, same process) at 91.0 / 105.3 ms per step against 77.6 --
    the tile shapes are chosen so that the whole batch fills whole rounds of the 256 CUs, and half batches do not."""
    from ullsam_amd import ops
    x32, pts, lbl, ids = inputs
    B = x32.shape[0]
    x = x32.to(dtype)  # pixel values as the model dtype sees them (app.py:522 moves pixels to the GPU in the model dtype)
    full = hasattr(model, "language_model")
    if full:
        am = torch.ones_like(ids)
        flags = (ids == 92546)[..., None].long()

    def compute():
        if full:
            out = model(pixel_values=x, input_ids=ids, attention_mask=am, image_flags=flags, return_dict=True, use_cache=False,
                        output_hidden_states=True)
            pe, md = model.prompt_encoder, model.mask_decoder
            sp, de = pe(points=(pts, lbl), boxes=None, masks=None, llm_hidden_states=out.hidden_states)
            low, iou = md(image_embeddings=out.image_embeddings, image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=sp,
                          dense_prompt_embeddings=de, multimask_output=False)
            _, mk = ops.resize_bilinear(low.contiguous(), (1024, 1024), want_float=False, threshold=0.0)
        else:  # configs[1]: Sam.forward, point prompt only
            recs = [{"image": x[b].float() * 255.0, "original_size": (1024, 1024), "point_coords": pts[b:b + 1], "point_labels": lbl[b:b + 1]}
                    for b in range(B)]
            outs = model(recs, multimask_output=False)
            low = torch.cat([o["low_res_logits"] for o in outs])
            mk = torch.cat([o["masks"] for o in outs]).to(torch.uint8)
        return low, mk

    return compute


def stub_compute(batch: int, rank: int):
    """`--stub` (control-path self-test, no GPU): a deterministic per-image CPU function with the real outputs' structure (fp32 low-res
    logits, u8 masks).  It exists so that the launcher / world assertion / barrier / max-over-ranks / gather / rank-0 JSON path of this
    file runs under gloo in the CPU tests; it measures nothing."""
    imgs = torch.from_numpy(np.random.default_rng(100 + rank).random((batch, 16, 16), dtype=np.float32))

    def compute():
        low = torch.sin(imgs * 3.0 + imgs.flip(-1)).reshape(batch, 1, 16, 16)
        return low, (low > 0.0).to(torch.uint8).repeat(1, 1, 4, 4)

    return compute


def make_step(compute, batch: int, world: int, gather: bool = True):
    """step() = compute() + (N > 1) the RCCL all-gather of its results, overlapped with the next step's compute."""
    from ullsam_amd import parallel
    pending = [None]

    def step():
        low, mk = compute()
        if world > 1 and gather:
            # the exchange of step k overlaps the compute of step k+1: at most one gather in flight
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = parallel.gather_mask_results_async(low, mk, None, counts=[batch] * world)
        return low, mk

    def drain():
        if pending[0] is not None:
            res = pending[0].wait()
            pending[0] = None
            return res
        return None

    step.drain = drain
    return step


def mask_iou_vs_fp32(model, vit, llm, inputs, device):
    """IoU of the bf16 masks of the timed configuration against the SAME weights and inputs run once in fp32 mode (exact-fp32 MFMA,
    the mode the parity tests pin to the reference within 1e-3): the `mask IoU vs ref` half of BASELINE.json's metric, on the
    bench configuration itself.  Outside the timed region."""
    from ullsam_amd import ops
    with torch.no_grad():
        low_b, mk_b = mask_path_compute(model, inputs, torch.bfloat16)()
        m32 = build_model(vit, llm, torch.float32, device, init=False)
        sd = {k: v.float() for k, v in model.state_dict().items()}
        missing, unexpected = m32.load_state_dict(sd, strict=False)
        assert not missing and not unexpected, (missing[:3], unexpected[:3])
        del sd
        x32, pts, lbl, ids = inputs
        low_f, mk_f = mask_path_compute(m32, (x32.to(torch.bfloat16).float(), pts, lbl, ids), torch.float32)()
        iou = ops.mask_iou(mk_b.contiguous(), mk_f.contiguous()).cpu().numpy()
        d = (low_b.float() - low_f.float()).abs().max().item()
        dm = (low_b.float() - low_f.float()).abs().mean().item()
        sc = low_f.float().abs().max().item()
        scm = low_f.float().abs().mean().item()
        frac = [float(m.float().mean().item()) for m in mk_f]
    del m32
    torch.cuda.empty_cache()
    return {"mean": round(float(iou.mean()), 6), "min": round(float(iou.min()), 6), "images": int(iou.size),
            "low_res_logit_max_abs_diff": round(d, 4), "low_res_logit_absmax": round(sc, 3),
            "low_res_logit_mean_abs_diff": round(dm, 5), "low_res_logit_mean_abs": round(scm, 4),
            "note": "the weights and (rank 0) tiles of tests/golden/full_depth.npz; see mask_iou_vs_reference for the score against the reference's own masks",
            "fp32_mask_fill_fraction": [round(f, 4) for f in frac],
            "reference": "same random weights + inputs through the fp32 mode of this library (pinned to the reference within 1e-3 / IoU delta < 1e-4 by tests/)"}


def mask_iou_vs_reference(model, inputs, dtype):
    """IoU of the timed configuration's masks against the REFERENCE's fp32 masks (tests/golden/full_depth.npz: the reference run at this
    depth on these weights, tiles, ids and clicks; oracle/gen_golden.py), next to the IoU the reference's own torch.autocast(bf16) reaches
    on the same tiles.  Rank 0, default workload only; outside the timed region."""
    path = os.path.join(ROOT, "tests", "golden", "full_depth.npz")
    if not os.path.exists(path):
        return {"skipped": "tests/golden/full_depth.npz not found"}
    from ullsam_amd import ops
    g = np.load(path)
    seeds = [int(v) for v in g["tile_seeds"]]
    n = min(len(seeds), inputs[0].shape[0])
    if seeds[:n] != list(FIXTURE_TILE_SEEDS[:n]):
        return {"skipped": "the fixture holds other tiles"}
    with torch.no_grad():
        low, mk = mask_path_compute(model, inputs, dtype)()
    ref = torch.from_numpy(np.stack([np.unpackbits(g[f"mask_bits_{i}"])[:1024 * 1024].reshape(1, 1024, 1024) for i in range(n)])).to(mk.device)
    iou = ops.mask_iou(mk[:n].contiguous(), ref.to(torch.uint8).contiguous()).cpu().numpy()
    dl = [float(np.abs(low[i].float().cpu().numpy() - g[f"low_{i}"][0]).mean()) for i in range(n)]
    return {"tile_seeds": seeds[:n], "iou": [round(float(v), 6) for v in iou], "mean": round(float(iou.mean()), 6), "min": round(float(iou.min()), 6),
            "reference_autocast_iou": [round(float(g[f"ac_mask_iou_{i}"]), 6) for i in range(n)],
            "reference_fp32_mask_fill": [round(float(g[f"mask_fill_{i}"]), 4) for i in range(n)],
            "low_res_logit_mean_abs_err": [round(v, 5) for v in dl],
            "reference_autocast_low_res_logit_mean_abs_err": [round(float(g[f"low_{i}_ac_mean_err"]), 5) for i in range(n)],
            "reference": "the reference (/root/reference, fp32, CPU) at full depth on the same weights / tiles / ids / clicks: tests/golden/full_depth.npz"}


def cpu_baseline(vit: str, llm: str, S: int, reps: int = 3):
    """The reference's algorithm (the oracle restated on torch CPU tensors for the two heavy stages, `kind: port-torch`; numpy oracle beside it) on the host cores, on a BOUNDED sample of the workload: one
    image; one windowed + one global ViT block, one LLM layer, the projectors and the whole prompt-encoder / mask-decoder /
    upsample tail are each timed `reps` times after one warm-up (median reported per stage) and the blocks / layers are
    extrapolated linearly in depth (profiles/r02_cpu_baseline_full_depth.json validates the extrapolation on a full-depth run)."""
    from oracle import ullsam_oracle as O
    v = VIT[vit]
    D, H = v["dim"], v["heads"]
    rng = np.random.default_rng(0)
    t_all = time.time()
    threads = None
    try:
        from threadpoolctl import threadpool_info
        threads = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
    except Exception:
        pass
    P = O.fill_state(O.vit_shapes(embed_dim=D, depth=2, num_heads=H, global_attn_indexes=(1,)), 0)
    xb = rng.standard_normal((1, 64, 64, D), dtype=np.float32)

    def timed(fn):
        fn()  # warm-up (BLAS thread pool, page faults)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return float(np.median(ts))

    # The two heavy stages (ViT blocks, LLM layers: 99 % of the time) run through oracle/torch_port.py = the oracle's arithmetic on torch CPU tensors, i.e. on ATen's
    # kernels, which is what the reference's own CPU path runs on (`kind: port-torch`); the same stages through the numpy oracle are timed once each beside them
    # (`numpy_port`: 3 - 5x slower -- chains of numpy temporaries -- and the only figure of rounds 1 - 4).
    from oracle import torch_port as TP
    PT = TP.to_torch(P)
    xt = torch.from_numpy(xb)
    # thread count: ATen's default (one thread per hardware thread) is NOT the fastest on a many-core host -- on the GPU box's 256 cores the 128-thread default ran an LLM layer
    # in 1.30 s, slower than 8 threads in the build container; a fair baseline uses the best of a short sweep (one timed run of the windowed block per candidate)
    default_threads = torch.get_num_threads()
    sweep = {}
    with torch.no_grad():
        for n in sorted({n for n in (8, 16, 32, 64, 128, default_threads) if n <= max(default_threads, 8)}):
            torch.set_num_threads(n)
            TP.vit_block(xt, PT, "blocks.0.", H, 14, 1e-6)
            t0 = time.perf_counter(); TP.vit_block(xt, PT, "blocks.0.", H, 14, 1e-6); sweep[n] = round(time.perf_counter() - t0, 4)
    torch_threads = min(sweep, key=sweep.get)
    torch.set_num_threads(torch_threads)
    with torch.no_grad():
        t_w = timed(lambda: TP.vit_block(xt, PT, "blocks.0.", H, 14, 1e-6))
        t_g = timed(lambda: TP.vit_block(xt, PT, "blocks.1.", H, 0, 1e-6))
    t0 = time.perf_counter(); O.vit_block(xb, P, "blocks.0.", H, 14, 1e-6); np_w = time.perf_counter() - t0
    t0 = time.perf_counter(); O.vit_block(xb, P, "blocks.1.", H, 0, 1e-6); np_g = time.perf_counter() - t0
    img = rng.random((1, 3, 1024, 1024), dtype=np.float32)
    t_fix = timed(lambda: O.vit_encoder(img, P, depth=0, num_heads=H, global_attn_indexes=()))
    n_g = len(v["glob"])
    t_vit = t_fix + (v["depth"] - n_g) * t_w + n_g * t_g
    t_llm = t_proj = t_layer = np_layer = 0.0
    if LLM[llm] is not None:
        c = LLM[llm]
        cfg = dict(hidden=c["hidden_size"], layers=1, heads=c["num_attention_heads"], kv_heads=c["num_key_value_heads"],
                   inter=c["intermediate_size"], vocab=8, rope_theta=1e6, eps=1e-5)
        PL = O.fill_state(O.internlm2_shapes(cfg["hidden"], 1, cfg["heads"], cfg["kv_heads"], cfg["inter"], 8, prefix="lm."), 0)
        emb = rng.standard_normal((1, S, cfg["hidden"]), dtype=np.float32)
        PLT, embt = TP.to_torch(PL), torch.from_numpy(emb)
        with torch.no_grad():
            t_layer = timed(lambda: TP.internlm2_layer(embt, PLT, "lm.model.layers.0.", cfg))
        t0 = time.perf_counter(); O.internlm2_model(PL, cfg, emb, prefix="lm."); np_layer = time.perf_counter() - t0
        t_llm = c["num_hidden_layers"] * t_layer
        PP = O.fill_state(O.projector_shapes(cfg["hidden"]), 0)
        feat = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
        hid = rng.standard_normal((1, 1024, cfg["hidden"]), dtype=np.float32)
        t_proj = timed(lambda: (O.extract_feature(PP, feat), O.text_aware_dense_feature(PP, hid)))
    PD = {}
    PD.update(O.fill_state(O.prompt_encoder_shapes(prefix="prompt_encoder."), 0))
    PD.update(O.fill_state(O.mask_decoder_shapes(prefix="mask_decoder."), 0))
    e = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)

    def dec():
        sp, de = O.prompt_encoder(PD, (np.array([[[500.0, 500.0]]], np.float32), np.array([[1]])), None, None, e, prefix="prompt_encoder.")
        low, _ = O.mask_decoder(PD, e, O.dense_pe(PD, prefix="prompt_encoder."), sp, de, False, prefix="mask_decoder.")
        return O.bilinear_resize(low, (1024, 1024)) > 0

    t_dec = timed(dec)
    total = t_vit + t_llm + t_proj + t_dec
    np_total = t_fix + (v["depth"] - n_g) * np_w + n_g * np_g + (LLM[llm]["num_hidden_layers"] * np_layer if LLM[llm] else 0.0) + t_proj + t_dec
    cores = os.cpu_count() or 1
    torch.set_num_threads(default_threads)
    return {"value": round(1.0 / total, 6), "unit": "images/s", "cores": torch_threads, "host_cores": cores, "blas_threads": threads, "torch_threads": torch_threads,
            "torch_thread_sweep_s": {str(k): v for k, v in sweep.items()},
            "kind": "port-torch", "reps": reps, "statistic": "median after 1 warm-up",
            "numpy_port": {"value": round(1.0 / np_total, 6), "vit_windowed_block_s": round(np_w, 4), "vit_global_block_s": round(np_g, 4), "llm_layer_s": round(np_layer, 4),
                           "note": "the same stages through the numpy oracle, one run each (the figure of rounds 1 - 4: kind 'port')"},
            "stages_s": {"vit_windowed_block": round(t_w, 4), "vit_global_block": round(t_g, 4), "vit_patch_embed_neck": round(t_fix, 4),
                         "llm_layer": round(t_layer, 4), "projectors_mlp1_mlp2": round(t_proj, 4), "prompt_mask_decoder_upsample": round(t_dec, 4),
                         "vit_total_extrapolated": round(t_vit, 2), "llm_total_extrapolated": round(t_llm, 2)},
            "sample": (f"torch-CPU restatement of the oracle (oracle/torch_port.py; patch-embed+neck, projectors and the decoder tail through the numpy oracle), fp32, 1 image: 1 windowed + 1 global ViT-{vit.upper()} block, patch-embed+neck, "
                       f"1 InternLM2-{llm} layer at S={S}, mlp1+mlp2, prompt-encoder+mask-decoder+upsample, each the median of {reps} runs; "
                       f"extrapolated linearly to {v['depth']} blocks / {LLM[llm]['num_hidden_layers'] if LLM[llm] else 0} layers; "
                       f"sampling took {time.time() - t_all:.0f}s")}


def decode_mode(a, device):
    """`--mode decode`: greedy decode throughput of the InternLM2-7B-shaped LLM (the caption path, app.py:431-495 ->
    modeling_internvl_sam.py:394-442): prefill S tokens, then K timed single-token steps at batch `--batch`.  A decode step streams
    every layer weight + the lm_head + the KV cache once, so the bound is HBM: achieved = those bytes / step time."""
    model = build_model("b", a.llm, torch.bfloat16, device)
    lm = model.language_model
    B, S, n = a.batch, a.seq, max(2, a.steps)
    ids = torch.randint(3, 90000, (B, S), device=device)

    def run(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = lm.generate(input_ids=ids, max_new_tokens=k, eos_token_id=-1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    for _ in range(max(1, a.warmup // 2)):
        run(4)
    # per-step time = (prefill + n + 1 tokens) - (prefill + 1 token), each the MINIMUM of three runs: the ~55 ms prefill inside both terms varies by
    # more than a decode step from run to run, and a single pair put that variation into the quotient (3.4 - 3.9 ms on one box)
    t1s = sorted(run(1)[0] for _ in range(3))
    tns = sorted(run(n + 1)[0] for _ in range(3))
    t1, tn = t1s[0], tns[0]
    per = (tn - t1) / n
    c = lm.config
    hd = c.hidden_size // c.num_attention_heads
    layer_w = ((c.num_attention_heads + 2 * c.num_key_value_heads) * hd * c.hidden_size + c.num_attention_heads * hd * c.hidden_size
               + 3 * c.intermediate_size * c.hidden_size)
    w_bytes = 2 * (c.num_hidden_layers * layer_w + c.vocab_size * c.hidden_size)
    kv_bytes = 2 * 2 * c.num_hidden_layers * B * c.num_key_value_heads * hd * (S + n // 2)
    ach = (w_bytes + kv_bytes) / per / 1e9
    line = {"metric": "greedy decode tokens/s (InternLM2-7B-shaped, bf16)", "value": round(B / per, 2), "unit": "tokens/s", "n_gpus": 1,
            "steps": n, "warmup": a.warmup, "ms_per_step": round(per * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random token ids, random-init weights)",
            "config": {"workload": f"InternVLSAMModel.generate's LLM loop: prefill {S} tokens then {n} greedy steps, batch {B}", "batch_per_gpu": B,
                       "seq_len": S, "prefill_plus_first_token_ms": round(t1 * 1e3, 1),
                       "runs_ms": {"prefill_plus_1": [round(t * 1e3, 2) for t in t1s], f"prefill_plus_{n + 1}": [round(t * 1e3, 2) for t in tns]}},
            "roofline": {"bound": "hbm", "kernel": "decode step (gemm_skinny_* weight streams + decode_attn_* over the KV cache)", "achieved": round(ach, 1),
                         "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                         "algorithmic_bytes_per_step": int(w_bytes + kv_bytes)}}
    print(json.dumps(line), flush=True)


def _spawn_ranks(n: int, argv) -> int:
    """WORLD_SIZE unset and --gpus N > 1: start the launcher as a CHILD process (this process never initialises the GPU) and
    relay its output; exit with its code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def traffic_record():
    """HBM-side bytes per GEMM launch: PMC counters cannot be read in-process, so the figure comes from the committed rocprofv3
    --pmc passes over this file (tools/collect_evidence.py), valid only for the kernel sources it was measured on."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_bench_traffic.json")))
    if not cands:
        return None, "no PMC record"
    tpath = cands[-1]   # the latest round's record
    rec = json.load(open(tpath))
    from ullsam_amd import build as _b
    if rec.get("csrc_digest") != _b._digest():
        return None, f"stale: {os.path.basename(tpath)} was measured on other kernel sources"
    return round(rec["hbm_bytes_per_launch"]), f"rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per GEMM launch, profiles/{os.path.basename(tpath)} (same kernel sources)"


def dist_setup(gpus: int, stub: bool):
    """Rank / world from the launcher's environment, the world-size assertion, the process group (RCCL; gloo in --stub) and every rank's
    device name.  -> (rank, world, device, ranks_seen)"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}; launch one rank per GPU (or leave WORLD_SIZE unset)")
    if stub:
        device, me = "cpu", f"rank {rank}: cpu (stub)"
    else:
        torch.cuda.set_device(local)
        device = f"cuda:{local}"
        me = f"rank {rank}: cuda:{local} {torch.cuda.get_device_name(local)}"
    ranks_seen = [me]
    if world > 1:
        import torch.distributed as dist
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))
        assert dist.get_world_size() == gpus
        names = [None] * world
        dist.all_gather_object(names, me)
        ranks_seen = names
    return rank, world, device, ranks_seen


def step_spread(ms):
    """min / median / max of the timed steps' own HIP-event spans (steps whose GEMM launches carry event pairs are listed apart: they run ~1.5 ms longer)."""
    if not ms:
        return None
    plain = [t for i, t in enumerate(ms) if i % GEMM_EVENT_EVERY != 0] or ms
    evs = [t for i, t in enumerate(ms) if i % GEMM_EVENT_EVERY == 0]
    srt = sorted(plain)
    return {"min": round(srt[0], 3), "median": round(srt[len(srt) // 2], 3), "max": round(srt[-1], 3), "n": len(srt),
            "median_of_steps_with_gemm_events": round(sorted(evs)[len(evs) // 2], 3) if evs else None}


def graph_replay_ms(compute, reps: int = 6):
    """The step captured ONCE in a HIP graph (every launch of the library goes to the current stream; tests/test_model_gpu.py::test_mask_path_replays_from_a_hip_graph
    pins that the replay reproduces the eager result) and replayed `reps` times behind one event pair -> ms per replay.  A diagnostic printed beside the eager `ms_per_step`
    (the headline stays the eager loop: at N > 1 a replay would overwrite buffers the all-gather of the previous step still reads)."""
    try:
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                compute()                                  # warm-up on the capture stream (lazy allocations, weight packing)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                compute()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / reps, 3), f"one HIP-graph capture of the step, {reps} replays behind one event pair, untimed; eager ms_per_step - this = what launching kernel by kernel costs"
    except Exception as e:                                 # a diagnostic must not take the measurement down
        return None, f"graph capture failed: {type(e).__name__}: {str(e)[:200]}"


def timed_steps(step, warmup: int, steps: int, world: int, device: str, on_timed=None, on_step=None):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; the last step's exchange is inside the
    timed region; -> (seconds = MAX over ranks, the last gathered result)."""
    gpu = device != "cpu"

    def barrier():
        if gpu:
            torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(warmup):
            step()
        step.drain()
        # The cyclic garbage collector is parked for the K timed steps (collected once before, re-enabled after): a generation-2 pass over a model of thousands of modules
        # lands in one step or another and showed as single steps of 91 / 105 ms among 76.5 ms ones (`step_ms.max`); nothing the step does is skipped
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        barrier()
        if on_timed:
            on_timed(True)
        marks = []                      # one HIP event per step boundary (K + 1 events: cheap) -> per-step times, for the spread printed beside the mean
        t0 = time.perf_counter()
        for i in range(steps):
            if on_step:
                on_step(i)
            if gpu:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
            step()
        if gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
        gathered = step.drain()
        barrier()
        dt = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        timed_steps.step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
        if on_timed:
            on_timed(False)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, gathered


def dist_teardown(world: int):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (BASELINE configs[2]: 4)")
    ap.add_argument("--vit", default="h", choices=list(VIT))
    ap.add_argument("--llm", default="7b", choices=list(LLM))
    ap.add_argument("--seq", type=int, default=1081)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-iou", action="store_true", help="skip the fp32 run that gives mask_iou_vs_fp32")
    ap.add_argument("--no-graph", action="store_true", help="skip the HIP-graph replay diagnostic (graph_ms_per_step): kernel traces then hold warm-up + timed steps only")
    ap.add_argument("--mode", default="mask", choices=["mask", "decode"], help="mask: the headline images/s path; decode: greedy tokens/s of the caption path")
    ap.add_argument("--stub", action="store_true", help="control-path self-test without a GPU (gloo, a stub step): measures nothing")
    ap.add_argument("--vit-fp8", action="store_true", help="fp8 (e4m3) operands for the ViT's LayerNorm-fed linears (qkv, lin1): BASELINE configs[4] 'fp8 MFMA ViT path'; everything else bf16")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(_spawn_ranks(a.gpus, sys.argv[1:]))
    rank, world, device, ranks_seen = dist_setup(a.gpus, a.stub)
    if a.stub:
        step = make_step(stub_compute(a.batch, rank), a.batch, world)
        dt, gathered = timed_steps(step, a.warmup, a.steps, world, device)
        if world > 1:
            assert gathered is not None and gathered[0].shape[0] == a.batch * world and gathered[1].shape[0] == a.batch * world
        if rank == 0:
            print(json.dumps({"metric": "STUB (bench.py control path self-test; not a measurement)", "value": round(a.batch * world * a.steps / dt, 2),
                              "unit": "stub steps", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
                              "stub": True, "data": "stub", "config": {"workload": "stub", "global_batch": a.batch * world, "ranks": ranks_seen},
                              "gathered_rows": None if gathered is None else int(gathered[0].shape[0])}), flush=True)
        dist_teardown(world)
        return
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.mode == "decode":
        if world != 1:
            raise SystemExit("bench.py --mode decode is a single-GPU measurement")
        with torch.no_grad():
            decode_mode(a, device)
        return

    model = build_model(a.vit, a.llm, dtype, device)
    if a.vit_fp8:
        if a.dtype != "bf16":
            raise SystemExit("bench.py --vit-fp8 needs --dtype bf16")
        (model.vision_model if hasattr(model, "vision_model") else model.image_encoder).fp8_linears = True
    if os.environ.get("ULLSAM_GEMM_VARIANT"):  # A/B switch for kernel experiments
        from ullsam_amd import _lib
        _lib.load().ullsam_set_gemm_variant(int(os.environ["ULLSAM_GEMM_VARIANT"]))
    timer = GemmTimer()
    full = LLM[a.llm] is not None
    inputs = make_inputs(a.batch, a.seq, device, full)
    step = make_step(mask_path_compute(model, inputs, dtype), a.batch, world)
    # The roofline's event pairs sit inside the timed region, and they are not free: an event pair around every one of the 269 GEMM launches of EVERY timed step cost
    # 1.5 ms of a 78 ms step (same box, same process order: 77.98 vs 76.45 ms per step with / without them).  They are therefore recorded on every GEMM_EVENT_EVERY-th timed
    # step only (steps 0, 5, 10, ...: >= 1000 timed launches at the default 20 steps); ULLSAM_BENCH_NO_GEMM_EVENTS=1 switches them off (A/B of their cost; `roofline` is then empty).
    ev_on = os.environ.get("ULLSAM_BENCH_NO_GEMM_EVENTS") != "1"
    dt, gathered = timed_steps(step, a.warmup, a.steps, world, device, on_timed=lambda on: setattr(timer, "on", False),
                               on_step=lambda i: setattr(timer, "on", ev_on and i % GEMM_EVENT_EVERY == 0))
    event_steps = len([i for i in range(a.steps) if i % GEMM_EVENT_EVERY == 0]) if ev_on else 0
    if world > 1:
        assert gathered is not None and gathered[0].shape[0] == a.batch * world and gathered[1].shape[0] == a.batch * world
    n_launch, gemm_ms, gemm_flops = timer.summary()
    graph_ms, graph_note = None, None
    if device != "cpu" and world == 1 and not a.no_graph:   # the same step ONCE MORE as a HIP-graph replay, outside the timed region: launch gaps of the eager loop show as eager - graph
        graph_ms, graph_note = graph_replay_ms(mask_path_compute(model, inputs, dtype))
    traffic, traffic_note = (None, "not the default workload")
    if a.vit == "h" and a.llm == "7b" and a.batch == 4 and a.dtype == "bf16":
        traffic, traffic_note = traffic_record()
    images = a.batch * world * a.steps
    value = images / dt
    peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else 157.3
    ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    line = {
        "metric": "images/s end-to-end (ViT+LLM+mask) 1024^2", "value": round(value, 4), "unit": "images/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": DATA_NOTE,
        "step_ms": step_spread(getattr(timed_steps, "step_ms", [])),
        "graph_ms_per_step": graph_ms, "graph_note": graph_note,
        "config": {"workload": (f"uLLSAM mask path (app.py:580-645): SAM ViT-{a.vit.upper()} + "
                                + (f"InternLM2-{a.llm}-shaped prefill S={a.seq} + " if full else "")
                                + f"prompt encoder + mask decoder + x4 upsample/threshold, 1 point prompt/image, batch {a.batch}/GPU"),
                   "batch_per_gpu": a.batch, "global_batch": a.batch * world, "seq_len": a.seq if full else 0,
                   "vit_linears": "fp8 e4m3 operands for qkv / lin1 (per-row / per-channel scales), proj / lin2 bf16" if a.vit_fp8 else "bf16",
                   "parallelism": f"dp{world} (images sharded, weights replicated, one RCCL all-gather of logits+masks per step, overlapped with the next step)",
                   "ranks": ranks_seen},
        "roofline": {"bound": "mfma", "kernel": "ullsam_gemm (every nn.Linear / conv-as-GEMM launch)", "achieved": round(ach, 2),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": traffic_note,
                     "algorithmic_bytes_per_launch": round(timer.alg_bytes / max(n_launch, 1)),
                     "launches_per_step": n_launch // max(event_steps, 1), "avg_launch_us": round(gemm_ms * 1e3 / max(n_launch, 1), 2),
                     "event_timed_steps": event_steps, "event_timed_launches": n_launch,
                     "gemm_share_of_step": round(gemm_ms / max(event_steps, 1) / (dt / a.steps * 1e3), 4)},
    }
    if rank == 0 and a.dtype == "bf16" and not a.no_iou and a.vit == "h" and a.llm == "7b" and a.seq == 1081:
        line["mask_iou_vs_reference"] = mask_iou_vs_reference(model, inputs, dtype)
    if rank == 0 and world == 1 and a.dtype == "bf16" and not a.no_iou:
        line["mask_iou_vs_fp32"] = mask_iou_vs_fp32(model, a.vit, a.llm, inputs, device)
    if a.vit_fp8:   # the GEMM timer wraps ops.gemm / gemm_qkv_rope only: the fp8 launches (ops.gemm_fp8) are outside `roofline`
        line["roofline"]["note"] = "bf16 GEMM launches only; the fp8 qkv / lin1 launches are not in this figure"
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.vit, a.llm, a.seq)
        print(json.dumps(line), flush=True)
    dist_teardown(world)


if __name__ == "__main__":
    main()
