"""Synthetic 1024x1024 "microscopy" tiles (BASELINE.json north_star: "throughput on synthetic 1024x1024 microscopy tiles").

No datasets exist offline, so the bench, the full-depth parity fixture and the tests share this numpy generator: flat-intensity
discs ("cells") on a dark background with a little sensor noise.  Why this and not uniform noise: a tile of i.i.d. pixels gives
every token an unrelated embedding and mask logits with a Gaussian marginal centred on the threshold, so the share of pixels
within the bf16 error of zero is ~0.8 * (error / sigma) whatever the weights are -- an IoU computed on such a mask measures the
input, not the arithmetic.  Cells give two populations of tokens (inside / outside), hence two modes of logits with a margin
between them, which is what a mask of a real microscopy image looks like.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def microscopy_tile(seed: int, size: int = 1024, n_cells: int = 14, r_range: Tuple[float, float] = (70.0, 150.0),
                    bg: float = 0.1, fg: float = 0.8, noise: float = 0.02) -> Tuple[np.ndarray, np.ndarray]:
    """-> (image float32 [3, size, size] in [0, 1], cell centres float32 [n_cells, 2] as (x, y) pixel coordinates).
    Deterministic in `seed` (numpy default_rng: the same tile on the GPU box and in the build container)."""
    rng = np.random.default_rng([int(seed), 0x5EED])
    cx = rng.uniform(0, size, n_cells)
    cy = rng.uniform(0, size, n_cells)
    r = rng.uniform(r_range[0], r_range[1], n_cells)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    inside = np.zeros((size, size), bool)
    for i in range(n_cells):
        inside |= (xx - np.float32(cx[i])) ** 2 + (yy - np.float32(cy[i])) ** 2 < np.float32(r[i]) ** 2
    img = np.where(inside, np.float32(fg), np.float32(bg)) + np.float32(noise) * rng.standard_normal((size, size), dtype=np.float32)
    img = np.clip(img, 0.0, 1.0).astype(np.float32)
    centres = np.stack([cx, cy], 1).astype(np.float32)
    return np.repeat(img[None], 3, 0), centres


def microscopy_batch(seeds, size: int = 1024, **kw) -> Tuple[np.ndarray, np.ndarray]:
    """-> (images [B, 3, size, size], one positive click per image [B, 1, 2]: the centre of the cell nearest the tile's middle)."""
    imgs, pts = [], []
    for s in seeds:
        im, c = microscopy_tile(int(s), size, **kw)
        k = int(np.argmin(((c - size / 2) ** 2).sum(1)))
        imgs.append(im)
        pts.append(np.clip(c[k], 0, size - 1))
    return np.stack(imgs), np.stack(pts)[:, None, :].astype(np.float32)


def param_init_rule(name: str, shape) -> Tuple[float, float]:
    """(mean, std) of the normal distribution bench.py draws parameter `name` from: the same per-name table the reference-generated
    fixtures were filled with (oracle/ullsam_oracle.py::fill_param; tests/test_host_cpu.py checks the two agree), so the bench runs on
    weights with the statistics the full-depth parity fixture pins.  fan-in scaling for matrices, unit-ish norm weights, small biases;
    parameters the reference initialises to zero (pos_embed, rel_pos_*, llm_bias) are non-zero so their code paths are live."""
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if "llm_scale_factor" in name:
        return 0.1, 0.02
    if "llm_bias" in name:
        return 0.05, 0.02
    if "positional_encoding_gaussian_matrix" in name:
        return 0.0, 1.0
    if "rel_pos" in name:
        return 0.0, 0.1
    if "pos_embed" in name:
        return 0.0, 0.05
    is_norm = ("norm" in name) or name.endswith(("neck.1.weight", "neck.3.weight", "neck.1.bias", "neck.3.bias")) \
        or name.startswith(("mlp1.0.", "mlp2.0.")) or ".mlp1.0." in name or ".mlp2.0." in name \
        or "output_upscaling.1." in name or "mask_downscaling.1." in name or "mask_downscaling.4." in name
    if is_norm:
        return (1.0, 0.1) if leaf == "weight" else (0.0, 0.05)
    if leaf == "bias" or len(shape) == 1:
        return 0.0, 0.05
    if any(k in name for k in ("tok_embeddings", "iou_token", "mask_tokens", "point_embeddings", "not_a_point_embed", "no_mask_embed")):
        return 0.0, 0.5
    if "output_upscaling" in name and len(shape) == 4:   # ConvTranspose2d [Cin, Cout, 2, 2]
        return 0.0, shape[0] ** -0.5
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return 0.0, max(fan_in, 1) ** -0.5


def fixture_param(name: str, shape, seed: int = 0) -> np.ndarray:
    """The VALUES the reference-generated parity fixtures were filled with (oracle/ullsam_oracle.py::fill_param, bit for bit: tests/test_host_cpu.py):
    an independent numpy stream per parameter name, `default_rng([seed, crc32(name)])`, standard normal draws mapped by the per-name rule above.
    bench.py fills its model with these (seed 0) so that its four default tiles are the tiles of tests/golden/full_depth.npz on the fixture's own
    weights, and its bf16 masks can be scored against the REFERENCE's fp32 masks stored there."""
    import math
    import zlib
    shape = tuple(int(s) for s in shape)
    x = np.random.default_rng([int(seed), zlib.crc32(name.encode())]).standard_normal(shape, dtype=np.float32)
    leaf = name.split(".")[-1]
    f32 = np.float32
    if "llm_scale_factor" in name:
        return (0.1 + 0.02 * x).astype(f32)
    if "llm_bias" in name:
        return (0.05 + 0.02 * x).astype(f32)
    if "positional_encoding_gaussian_matrix" in name:
        return x
    if "rel_pos" in name:
        return (0.1 * x).astype(f32)
    if "pos_embed" in name:
        return (0.05 * x).astype(f32)
    is_norm = ("norm" in name) or name.endswith(("neck.1.weight", "neck.3.weight", "neck.1.bias", "neck.3.bias")) \
        or name.startswith(("mlp1.0.", "mlp2.0.")) or ".mlp1.0." in name or ".mlp2.0." in name \
        or "output_upscaling.1." in name or "mask_downscaling.1." in name or "mask_downscaling.4." in name
    if is_norm:
        return (1.0 + 0.1 * x).astype(f32) if leaf == "weight" else (0.05 * x).astype(f32)
    if leaf == "bias" or len(shape) == 1:
        return (0.05 * x).astype(f32)
    if any(k in name for k in ("tok_embeddings", "iou_token", "mask_tokens", "point_embeddings", "not_a_point_embed", "no_mask_embed")):
        return (0.5 * x).astype(f32)
    if "output_upscaling" in name and len(shape) == 4:   # ConvTranspose2d [Cin, Cout, 2, 2]
        return (x / math.sqrt(shape[0])).astype(f32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return (x / math.sqrt(max(fan_in, 1))).astype(f32)


def fill_model_like_fixtures(model, seed: int = 0, workers: int = 32):
    """Every floating-point parameter / persistent buffer of `model` <- fixture_param(name) (produced on a thread pool -- numpy's generators
    release the GIL -- and copied straight into the device tensors).  Sam's pixel_mean / pixel_std (constants, not weights) are left alone."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    sd = {k: v for k, v in model.state_dict().items() if v.is_floating_point() and "pixel_mean" not in k and "pixel_std" not in k}

    def one(kv):
        k, v = kv
        v.copy_(torch.from_numpy(fixture_param(k, tuple(v.shape), seed)).to(v.dtype))

    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(one, sd.items()))
    return model


def blob_decoder_state(gaussian_matrix: np.ndarray, beta2: float = 3.0, taus=(0.62, 0.72, 0.80, 0.86), gamma: float = 24.0) -> dict:
    """A STRUCTURED (not random, not trained) parameter set for SAM's prompt encoder + mask decoder under which a positive click draws a disc
    around itself -- used by the automatic-mask-generator test and bench (BASELINE configs[4]), where a random decoder draws full-frame
    textures whose boxes are all the whole tile, so box NMS would have nothing to do.  No weights exist offline; this one is written down.

    Mechanism (SAM's own): in layer 0's image -> token attention a pixel p scores the click token by <PE(p), PE(c)> = sum_j cos(2 pi g_j.(p - c)),
    a kernel peaked at the click (random Fourier features, prompt_encoder.py:206-250), against null tokens whose keys are zero; the click token's
    value writes the attention weight into one reserved channel of the image tokens.  Everything else is switched off exactly (zero out_proj /
    lin2), so the channel survives to the upscaling and the hypernetwork product as  logit = gamma (F(m(p)) - F(tau_i)),  m = the mean over the
    8 heads of the click's attention weight, F monotone.  The four mask tokens differ in tau: discs of four radii per click.
    Returns {state_dict key (relative to the Sam model): float32 array} for the keys it sets; all other parameters keep their values."""
    G = np.asarray(gaussian_matrix, np.float32)
    C = 2 * G.shape[1]
    assert C == 256, "laid out for SAM's 256-channel decoder"
    r_click, r_pad, r_out, ch_a, ch_b = 0, 1, 2, 3, 128     # reserved channels: token identities, the blob channel, its constant partner
    B, A, D = 30.0, 4.0, 2.0
    z = lambda *s: np.zeros(s, np.float32)
    sd = {}

    def onehot(ch, v):
        x = z(1, C)
        x[0, ch] = v
        return x

    # prompt encoder: token identities on reserved channels, no dense prompt
    sd["prompt_encoder.point_embeddings.1.weight"] = onehot(r_click, B)
    sd["prompt_encoder.point_embeddings.0.weight"] = onehot(r_pad, B)
    sd["prompt_encoder.not_a_point_embed.weight"] = onehot(r_pad, B)
    sd["prompt_encoder.no_mask_embed.weight"] = z(1, C)
    sd["mask_decoder.iou_token.weight"] = onehot(r_out, B)
    sd["mask_decoder.mask_tokens.weight"] = np.repeat(onehot(r_out, B), 4, 0)
    t = "mask_decoder.transformer."

    def zero_out(prefix, internal):
        sd[prefix + "out_proj.weight"] = z(C, internal)
        sd[prefix + "out_proj.bias"] = z(C)

    for li in range(2):
        b = f"{t}layers.{li}."
        for n in ("norm1", "norm2", "norm3", "norm4"):
            sd[b + n + ".weight"] = np.ones(C, np.float32)
            sd[b + n + ".bias"] = z(C)
        sd[b + "mlp.lin2.weight"] = z(C, 2048)
        sd[b + "mlp.lin2.bias"] = z(C)
        zero_out(b + "cross_attn_token_to_image.", C // 2)
        if li == 1:
            zero_out(b + "self_attn.", C)
            zero_out(b + "cross_attn_image_to_token.", C // 2)
    zero_out(t + "final_attn_token_to_image.", C // 2)
    sd[t + "norm_final_attn.weight"] = np.ones(C, np.float32)
    sd[t + "norm_final_attn.bias"] = z(C)
    # layer 0 self-attention (no residual, transformer.py:157-158): every token attends to the tokens of its own reserved channel and copies that channel
    b = f"{t}layers.0.self_attn."
    eye3 = z(C, C)
    for ch in (r_click, r_pad, r_out):
        eye3[ch, ch] = 1.0
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
        sd[b + n + ".weight"] = eye3.copy()
        sd[b + n + ".bias"] = z(C)
    # layer 0 image -> token attention: head h scores with 8 frequencies (sin and cos rows of the positional encoding)
    b = f"{t}layers.0.cross_attn_image_to_token."
    Wq = z(C // 2, C)
    # the 64 LOWEST frequencies |g_f| among those whose sin row is not a reserved channel, dealt round-robin to the heads: a wide kernel (a disc of
    # ~1/6 of the image side) that every head sees at the same scale
    order = [int(f) for f in np.argsort((G.astype(np.float64) ** 2).sum(0)) if f not in (r_click, r_pad, r_out, ch_a)][:64]
    for n, f in enumerate(order):
        h, i = n % 8, n // 8
        Wq[16 * h + i, f] = 1.0                    # sin(2 pi g_f . x)
        Wq[16 * h + 8 + i, C // 2 + f] = 1.0       # cos(2 pi g_f . x)
    s = float(np.sqrt(beta2))
    sd[b + "q_proj.weight"] = (s * Wq).astype(np.float32)
    sd[b + "k_proj.weight"] = (s * Wq).astype(np.float32)
    sd[b + "q_proj.bias"] = z(C // 2)
    sd[b + "k_proj.bias"] = z(C // 2)
    Wv = z(C // 2, C)
    Wo = z(C, C // 2)
    for h in range(8):
        Wv[16 * h, r_click] = 1.0 / 16.0           # the click token's queries are ~16 e_click after the LayerNorms; null tokens carry ~0 there
        Wo[ch_a, 16 * h] = A / 8.0
    sd[b + "v_proj.weight"] = Wv
    sd[b + "v_proj.bias"] = z(C // 2)
    sd[b + "out_proj.weight"] = Wo
    ob = z(C)
    ob[ch_b] = D
    sd[b + "out_proj.bias"] = ob
    # upscaling: channels a / b ride through both transposed convolutions as a nearest-neighbour x4 (LayerNorm2d sees the pair, so the ratio survives)
    w0 = z(C, C // 4, 2, 2)
    w0[ch_a, 0] = 1.0
    w0[ch_b, 1] = 1.0
    sd["mask_decoder.output_upscaling.0.weight"] = w0
    sd["mask_decoder.output_upscaling.0.bias"] = z(C // 4)
    sd["mask_decoder.output_upscaling.1.weight"] = np.ones(C // 4, np.float32)
    sd["mask_decoder.output_upscaling.1.bias"] = z(C // 4)
    w1 = z(C // 4, C // 8, 2, 2)
    w1[0, 0] = 1.0
    sd["mask_decoder.output_upscaling.3.weight"] = w1
    b1 = z(C // 8)
    b1[1] = 4.0                                    # a constant channel (GELU(4) = 4.0): carries the thresholds
    sd["mask_decoder.output_upscaling.3.bias"] = b1

    def F(m):                                      # what the two LayerNorms make of the attention weight m (channel a: A m, channel b: D)
        x = A * m
        return 8.0 * x / np.sqrt(x * x + D * D)

    gelu4 = 4.0 * 0.5 * (1.0 + math.erf(4.0 / math.sqrt(2.0)))
    for i in range(4):
        p = f"mask_decoder.output_hypernetworks_mlps.{i}.layers."
        sd[p + "2.weight"] = z(C // 8, C)
        hb = z(C // 8)
        hb[0] = gamma
        hb[1] = -gamma * F(taus[i]) / gelu4
        sd[p + "2.bias"] = hb
    sd["mask_decoder.iou_prediction_head.layers.2.weight"] = z(4, 256)
    sd["mask_decoder.iou_prediction_head.layers.2.bias"] = np.asarray([0.95, 0.93, 0.91, 0.89], np.float32)
    return sd


def blob_decoder_init(sam, neck_gain: float = 1e-3, **kw):
    """Load blob_decoder_state into a Sam model (ullsam_amd or reference layout) and shrink the image encoder's last LayerNorm2d so that the (random)
    image embedding does not disturb the positional scores.  In place; returns the model."""
    import torch
    G = sam.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.detach().float().cpu().numpy()
    sd = blob_decoder_state(G, **kw)
    own = sam.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            own[k].copy_(torch.from_numpy(v).to(own[k].dtype))
        own["image_encoder.neck.3.weight"].fill_(neck_gain)
        own["image_encoder.neck.3.bias"].zero_()
    return sam
