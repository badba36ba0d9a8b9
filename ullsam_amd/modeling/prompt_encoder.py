"""SAM prompt encoder (+ uLLSAM's LLM-conditioned dense prompt) on HIP kernels.

API mirror of modeling/prompt_encoder.py: PromptEncoder(embed_dim, image_embedding_size, input_image_size,
mask_in_chans, activation); forward(points, boxes, masks, llm_hidden_states=None) -> (sparse[P,n,C] fp32,
dense[P,C,H,W] fp32); get_dense_pe() -> [1,C,H,W].  Same state_dict keys.
Sparse embeddings are fp32 even for bf16 models, as in the reference (prompt_encoder.py:178-182).
"""
from __future__ import annotations

from typing import Optional, Tuple, Type

import torch
from torch import nn

from .. import ops
from .common import LayerNorm2d, Packed


class _Embedding(Packed):
    def __init__(self, n: int, dim: int):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, dim))


class _Conv(Packed):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k) * (cin * k * k) ** -0.5)
        self.bias = nn.Parameter(torch.zeros(cout))


class PositionEmbeddingRandom(Packed):
    """prompt_encoder.py:206-250."""

    def __init__(self, num_pos_feats: int = 64, scale: Optional[float] = None) -> None:
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))

    def G(self) -> torch.Tensor:
        return self.f32("G", self.positional_encoding_gaussian_matrix)

    def forward(self, size: Tuple[int, int]) -> torch.Tensor:
        """C x H x W positional encoding grid."""
        h, w = size
        pe = ops.dense_pe(self.G(), h, w)  # [h*w, C]
        return ops.transpose(pe.reshape(1, h * w, -1), 1, h * w, pe.shape[-1]).reshape(-1, h, w)


class PromptEncoder(Packed):
    def __init__(self, embed_dim: int, image_embedding_size: Tuple[int, int], input_image_size: Tuple[int, int],
                 mask_in_chans: int, activation: Type[nn.Module] = nn.GELU) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.input_image_size = input_image_size
        self.image_embedding_size = image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings: int = 4
        self.point_embeddings = nn.ModuleList([_Embedding(1, embed_dim) for _ in range(self.num_point_embeddings)])
        self.not_a_point_embed = _Embedding(1, embed_dim)
        self.llm_scale_factor = nn.Parameter(torch.ones(1) * 0.1)  # prompt_encoder.py:50
        self.llm_bias = nn.Parameter(torch.zeros(1))               # prompt_encoder.py:51
        self.mask_input_size = (4 * image_embedding_size[0], 4 * image_embedding_size[1])
        if activation is not nn.GELU:
            raise NotImplementedError("mask_downscaling is fused with GELU (the only activation SAM builds)")
        self.mask_downscaling = nn.Sequential(_Conv(1, mask_in_chans // 4, 2), LayerNorm2d(mask_in_chans // 4), activation(),
                                              _Conv(mask_in_chans // 4, mask_in_chans, 2), LayerNorm2d(mask_in_chans), activation(),
                                              _Conv(mask_in_chans, embed_dim, 1))
        self.no_mask_embed = _Embedding(1, embed_dim)

    # -- token-major (NHWC) internals -----------------------------------------------------------
    def dense_pe_tokens(self) -> torch.Tensor:
        h, w = self.image_embedding_size
        G = self.pe_layer.G()
        return self.pk("dense_pe", G, lambda: ops.dense_pe(G, h, w))  # constant: cached

    def get_dense_pe(self) -> torch.Tensor:
        h, w = self.image_embedding_size
        pe = self.dense_pe_tokens()
        return ops.transpose(pe.reshape(1, h * w, -1), 1, h * w, self.embed_dim).reshape(1, self.embed_dim, h, w)

    def _emb_table(self) -> torch.Tensor:
        srcs = [self.not_a_point_embed.weight] + [e.weight for e in self.point_embeddings]
        return self.pk("emb_table", srcs, lambda: torch.cat([s.detach().float() for s in srcs], 0).contiguous())

    def sparse_tokens(self, points, boxes) -> torch.Tensor:
        dev = self.no_mask_embed.weight.device
        if points is None and boxes is None:
            return torch.empty((1, 0, self.embed_dim), device=dev)
        coords = labels = None
        P, Np = 0, 0
        if points is not None:
            coords = points[0].to(dev).float().contiguous()
            labels = points[1].to(dev).to(torch.int32).contiguous()
            P, Np = coords.shape[0], coords.shape[1]
        bx = None
        if boxes is not None:
            bx = boxes.to(dev).float().reshape(-1, 4).contiguous()
            P = bx.shape[0] if points is None else P
        pad = 1 if (points is not None and boxes is None) else 0
        return ops.sparse_embed(coords, labels, bx, self.pe_layer.G(), self._emb_table(), P, Np, pad, self.embed_dim,
                                self.input_image_size[1], self.input_image_size[0])

    def dense_tokens(self, bs: int, masks, llm_hidden_tokens: Optional[torch.Tensor]) -> torch.Tensor:
        """Dense prompt, token-major fp32 [bs or 1, H*W, C] (a single row set broadcasts over prompts)."""
        h, w = self.image_embedding_size
        if masks is not None:
            md = self.mask_downscaling
            prm = [md[0].weight, md[0].bias, md[1].weight, md[1].bias, md[3].weight, md[3].bias, md[4].weight, md[4].bias,
                   md[6].weight, md[6].bias]
            prm = [self.f32(f"md{i}", t) for i, t in enumerate(prm)]
            return ops.mask_downscale(masks.float().contiguous(), h, w, self.embed_dim, prm)
        if llm_hidden_tokens is not None:
            # _normalize_llm_hidden_states, prompt_encoder.py:131-151: per-pixel LN over C (no affine, eps 1e-5) * scale + bias
            return ops.norm(llm_hidden_tokens, None, None, 1e-5, torch.float32, post_scale=self.f32("ls", self.llm_scale_factor),
                            post_shift=self.f32("lb", self.llm_bias))
        return self.f32("nme", self.no_mask_embed.weight).reshape(1, 1, self.embed_dim)

    # -- reference API ------------------------------------------------------------------------------
    def _get_batch_size(self, points, boxes, masks) -> int:
        if points is not None:
            return points[0].shape[0]
        if boxes is not None:
            return boxes.shape[0]
        if masks is not None:
            return masks.shape[0]
        return 1

    def forward(self, points, boxes, masks, llm_hidden_states: Optional[torch.Tensor] = None):
        from .. import training
        if training.wants_autograd(self, llm_hidden_states):   # train() mode with gradients on (train_joint_v2.py:1055-1060)
            return training.prompt_encoder_forward(self, points, boxes, masks, llm_hidden_states)
        with torch.no_grad():
            return self._forward_inference(points, boxes, masks, llm_hidden_states)

    def _forward_inference(self, points, boxes, masks, llm_hidden_states: Optional[torch.Tensor] = None):
        bs = self._get_batch_size(points, boxes, masks)
        h, w = self.image_embedding_size
        C = self.embed_dim
        sparse = self.sparse_tokens(points, boxes)
        if sparse.shape[1] == 0:
            sparse = sparse.expand(bs, 0, C)
        llm_tok = None
        if masks is None and llm_hidden_states is not None:
            x = llm_hidden_states.float().contiguous()
            llm_tok = ops.transpose(x.reshape(x.shape[0], C, h * w), x.shape[0], C, h * w)
        dt = self.dense_tokens(bs, masks, llm_tok)
        if dt.shape[1] == 1:  # no_mask_embed broadcast (prompt_encoder.py:199-201)
            dense = dt.reshape(1, C, 1, 1).expand(bs, C, h, w)
        else:
            n = dt.shape[0]
            dense = ops.transpose(dt.reshape(n, h * w, C), n, h * w, C).reshape(n, C, h, w)
            if n != bs:
                dense = dense.reshape(bs, -1, h, w)  # same failure mode as the reference's reshape (:195-197)
        return sparse, dense
