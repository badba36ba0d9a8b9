"""SAM mask decoder on HIP kernels (API + state_dict mirror of modeling/mask_decoder.py)."""
from __future__ import annotations

from typing import List, Optional, Tuple, Type

import os

import torch
from torch import nn

from .. import ops
from ..packing import pack_convT_k2s2
from .common import LayerNorm2d, Linear, Packed


class _Embedding(Packed):
    def __init__(self, n, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, dim))


class _ConvT(Packed):
    """Parameter holder with nn.ConvTranspose2d(k=2, s=2)'s layout: weight [Cin, Cout, 2, 2]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cin, cout, 2, 2) * cin ** -0.5)
        self.bias = nn.Parameter(torch.zeros(cout))

    def packed(self, dt):
        w = self.pk("w", (self.weight, self.bias), lambda: pack_convT_k2s2(self.weight.detach(), self.bias.detach())[0].to(dt).contiguous())
        b = self.pk("b", (self.weight, self.bias), lambda: pack_convT_k2s2(self.weight.detach(), self.bias.detach())[1].float().contiguous())
        return w, b


class MLP(Packed):
    """mask_decoder.py:154-176 (ReLU between layers)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, sigmoid_output: bool = False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.float().contiguous()
        for i, l in enumerate(self.layers):
            x = l.tok(x, ops.ACT_RELU if i < self.num_layers - 1 else ops.ACT_NONE)
        if self.sigmoid_output:
            x = torch.sigmoid(x)
        return x


FUSED_HEADS = os.environ.get("ULLSAM_FUSED_HEADS", "1") != "0"   # bf16: hypernetwork MLPs + IoU head as one launch (csrc/dectok.hip)
FUSED_UP2 = os.environ.get("ULLSAM_FUSED_UP2", "1") != "0"   # bf16: second transposed convolution + GELU + hypernetwork product as one kernel (any prompt count)


class MaskDecoder(Packed):
    def __init__(self, *, transformer_dim: int, transformer: nn.Module, num_multimask_outputs: int = 3,
                 activation: Type[nn.Module] = nn.GELU, iou_head_depth: int = 3, iou_head_hidden_dim: int = 256) -> None:
        super().__init__()
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = _Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = _Embedding(self.num_mask_tokens, transformer_dim)
        if activation is not nn.GELU:
            raise NotImplementedError("output_upscaling is fused with GELU (the only activation SAM builds)")
        self.output_upscaling = nn.Sequential(_ConvT(transformer_dim, transformer_dim // 4), LayerNorm2d(transformer_dim // 4),
                                              activation(), _ConvT(transformer_dim // 4, transformer_dim // 8), activation())
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth)

    # -- token-major internals ----------------------------------------------------------------------
    def predict_masks_tokens(self, image_tokens: torch.Tensor, pe_tokens: torch.Tensor, sparse: torch.Tensor,
                             dense_tokens: torch.Tensor, hw: Tuple[int, int], image_cache: Optional[dict] = None,
                             mask_range: Optional[Tuple[int, int]] = None):
        """image_tokens fp32 [1 or P, N, C]; pe_tokens [N, C]; sparse fp32 [P, n, C]; dense_tokens fp32 [P|1, N|1, C].
        Returns (masks [P, 4, 4h, 4w] fp32, iou [P, 4] fp32) -- predict_masks, mask_decoder.py:112-149.
        mask_range (m0, m1): only masks m0 .. m1 - 1 and their IoU predictions are produced (forward's multimask slice, :100-105, taken BEFORE the hypernetwork product
        and the second upscaling instead of after them: the automatic mask generator keeps masks 1 .. 3 of every prompt)."""
        h, w = hw
        N, C = h * w, self.transformer_dim
        P = sparse.shape[0]
        dt = self.transformer.compute_dtype
        out_tok = self.pk("out_tokens", (self.iou_token.weight, self.mask_tokens.weight),
                          lambda: torch.cat([self.iou_token.weight.detach().float(), self.mask_tokens.weight.detach().float()], 0))
        tokens = ops.concat_token_rows(out_tok, sparse.float().contiguous())   # [iou token, mask tokens, sparse prompts] per prompt (:119-123), one launch
        # src = repeat_interleave(image_embeddings, P) + dense  (:126-127); row-modular broadcast of both operands
        dense_rows = dense_tokens.numel() // C
        Pk = 1 if (image_tokens.shape[0] == 1 and dense_rows in (1, N)) else P   # one image, prompt-independent dense embedding
        # image_cache (a dict owned by the caller, e.g. one per crop of the automatic mask generator): with ONE image and a prompt-independent dense
        # embedding the image side is the same for every batch of prompts until the first image -> token attention -- keys, their model-dtype copies
        # and layer 0's K / V projections are computed by the first call and reused by the next ones (transformer.py:220-242 re-runs them per call).
        cache = image_cache if (image_cache is not None and Pk == 1) else None
        if cache is not None:
            # what the cached tensors were computed FROM: another image / dense embedding (or an in-place update of the same storage), another dtype or token
            # count empties the dict instead of serving the previous image's keys (weights are the caller's side of the contract: a dict lives for one crop)
            # The dict also KEEPS the two source tensors: their storage cannot be freed and handed to another image at the same address while the cache lives, and
            # identity (`is`) + version tells "the same embedding, unchanged".  Inference tensors (torch.inference_mode) have no version counter: version None.
            def _ver(t):
                try:
                    return t._version
                except RuntimeError:
                    return None
            sig = (_ver(image_tokens), tuple(image_tokens.shape), _ver(dense_tokens), tuple(dense_tokens.shape), str(dt), N)
            src = cache.get("_source")
            if src is None or src[0] is not image_tokens or src[1] is not dense_tokens or src[2] != sig:
                cache.clear()
                cache["_source"] = (image_tokens, dense_tokens, sig)
        if cache is not None and "keys" in cache:
            keys = cache["keys"]
        else:
            keys = ops.add_cast(image_tokens.reshape(-1, C), dense_tokens.reshape(-1, C).contiguous(), torch.float32, rows=Pk * N)
            if cache is not None:
                cache["keys"] = keys
        hs, src = self.transformer.forward_tokens(keys.reshape(Pk, N, C), pe_tokens, tokens, keys_in_compute_dtype=True, cache=cache)
        m0, m1 = mask_range if mask_range is not None else (0, self.num_mask_tokens)
        nm = m1 - m0
        up0, ln, up1 = self.output_upscaling[0], self.output_upscaling[1], self.output_upscaling[3]
        w0, b0 = up0.packed(dt)
        w1, b1 = up1.packed(dt)
        c4, c8 = C // 4, C // 8
        if FUSED_UP2 and dt == torch.bfloat16 and C == 256:
            u1 = ops.up1_ln_gelu(src.reshape(P * N, C), w0, b0, *ln.wb(), ln.eps)        # first transposed convolution + LayerNorm2d + GELU: the fp32 convolution output is never written
        else:
            u1 = ops.gemm(src.reshape(P * N, C), w0, b0, out_f32=True)           # [P*N, (ky,kx,c4)]
            u1 = ops.norm(u1.reshape(P * N * 4, c4), *ln.wb(), ln.eps, dt, act=ops.ACT_GELU)   # LayerNorm2d + GELU per output pixel
        heads = self._fused_heads(dt, T_all=hs.shape[1]) if (FUSED_HEADS and dt == torch.bfloat16) else None
        if heads is not None:                                                # the four hypernetwork MLPs + the IoU head: 15 token-side linears, one launch
            hyper, iou_all = ops.dec_heads(hs.contiguous(), heads[0], heads[1], P, hs.shape[1], self.num_mask_tokens, m0, nm)
        else:
            hyper = torch.stack([self.output_hypernetworks_mlps[i](hs[:, 1 + i, :]) for i in range(m0, m1)], dim=1).contiguous()
        if FUSED_UP2 and dt == torch.bfloat16 and c4 == 64 and c8 == 32 and nm <= 8:
            masks = ops.up2_hyper_masks(u1, w1, b1, hyper, P, nm, h, w)      # second transposed convolution + GELU + hypernetwork product: the upscaled embedding is never written
        else:
            u2 = ops.gemm(u1, w1, b1, act=ops.ACT_GELU)                      # [P*N*4, (ky2,kx2,c8)]
            masks = ops.hyper_masks(u2, hyper, P, nm, h, w, c8)
        iou = iou_all if heads is not None else self.iou_prediction_head(hs[:, 0, :])
        return masks, (iou if mask_range is None else iou[:, m0:m1])

    def _fused_heads(self, dt, T_all: int):
        """Host arrays of the 15 weight / bias device pointers ullsam_dec_heads takes (chain-major: hypernetwork MLP 0 .. 3, IoU head), or None when the heads
        do not have SAM's shape (four mask tokens, three layers of width 256, 32 / <= 16 outputs, no sigmoid).  A chain's last weight is zero-padded to a
        multiple of 16 rows once per weight version (the pack cache keeps the padded copies -- and through them the pointers -- alive)."""
        C = self.transformer_dim
        chains = list(self.output_hypernetworks_mlps) + [self.iou_prediction_head]
        if self.num_mask_tokens != 4 or C != 256 or T_all < 5 or any(
                m.num_layers != 3 or m.sigmoid_output or m.layers[0].out_features != 256 or m.layers[1].out_features != 256 for m in chains):
            return None
        if any(m.layers[2].out_features != 32 for m in chains[:4]) or chains[4].layers[2].out_features != 4:
            return None
        ws, bs, keep = [], [], []
        for ci, m in enumerate(chains):
            for li, l in enumerate(m.layers):
                w = l.pk("w:mfma_rows", l.weight, lambda l=l: ops.pack_mfma_rows(l.weight))    # fragment order, zero-padded to a multiple of 16 rows
                b = l.b()
                keep += [w, b]
                ws.append(w.data_ptr())
                bs.append(0 if b is None else b.data_ptr())
        key = (tuple(ws), tuple(bs))
        hit = getattr(self, "_heads_ptrs", None)
        if hit is None or hit[0] != key:
            hit = (key, torch.tensor(ws, dtype=torch.int64), torch.tensor(bs, dtype=torch.int64), keep)
            object.__setattr__(self, "_heads_ptrs", hit)
        return hit[1], hit[2]

    # -- reference API -------------------------------------------------------------------------------
    @torch.no_grad()
    def predict_masks(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings):
        b, c, h, w = image_embeddings.shape
        P = sparse_prompt_embeddings.shape[0]
        tok = lambda x, n: ops.transpose(x.float().contiguous().reshape(n, c, h * w), n, c, h * w)
        img = tok(image_embeddings, b)
        pe = tok(image_pe[:1], 1).reshape(h * w, c)
        d = dense_prompt_embeddings
        if d.shape[0] != P or d.stride(-1) == 0 or d.stride(-2) == 0:  # broadcast views (no_mask_embed.expand)
            d = d.expand(P, c, h, w)
        dense = tok(d, P)
        if b != 1 and b != P:
            raise ValueError(f"image_embeddings batch {b} is incompatible with {P} prompts (reference repeat_interleave semantics)")
        return self.predict_masks_tokens(img, pe, sparse_prompt_embeddings, dense, (h, w))

    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output: bool):
        from .. import training
        if training.wants_autograd(self, image_embeddings, sparse_prompt_embeddings, dense_prompt_embeddings):   # train_joint_v2.py:1063-1069
            return training.mask_decoder_forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output)
        with torch.no_grad():
            masks, iou_pred = self.predict_masks(image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings)
            sl = slice(1, None) if multimask_output else slice(0, 1)  # mask_decoder.py:100-105
            return masks[:, sl, :, :], iou_pred[:, sl]
