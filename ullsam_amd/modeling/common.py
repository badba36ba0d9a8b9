"""Parameter holders shared by the SAM modules + the weight-pack cache.

Mirrors modeling/common.py of the reference (MLPBlock :13-26, LayerNorm2d :31-43) at the parameter level: same
attribute names => same state_dict keys.  Compute goes through ullsam_amd.ops (HIP kernels).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple, Type

import torch
from torch import nn

from .. import ops


class PackCache:
    """Caches kernel-side re-layouts of parameters (dtype casts, im2col weight orders, fp32 copies of biases).

    Keyed by (name) and validated against the source tensors' (data_ptr, _version, dtype, device), so .to(),
    load_state_dict() and in-place edits all invalidate the pack."""

    def __init__(self):
        self._store: Dict[str, Tuple[tuple, torch.Tensor]] = {}

    def get(self, name: str, srcs, fn: Callable[[], torch.Tensor]) -> torch.Tensor:
        if isinstance(srcs, torch.Tensor):
            srcs = (srcs,)
        key = tuple((s.data_ptr(), s._version, s.dtype, str(s.device)) for s in srcs)
        hit = self._store.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            val = fn()
        self._store[name] = (key, val)
        return val


class Packed(nn.Module):
    """nn.Module with a pack cache that is never part of the state_dict."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_packs", PackCache())

    def pk(self, name, srcs, fn):
        return self._packs.get(name, srcs, fn)

    def f32(self, name: str, p: torch.Tensor) -> torch.Tensor:
        """fp32 contiguous view/copy of a parameter (biases, norm weights, fp32-only tables)."""
        return self.pk(name + ":f32", p, lambda: p.detach().float().contiguous())

    def cdt(self, name: str, p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """Parameter in the compute dtype, contiguous (no copy when it already is)."""
        return self.pk(name + ":cdt", p, lambda: p.detach().to(dtype).contiguous())


class Linear(Packed):
    """Parameter holder with nn.Linear's state_dict layout (weight [out, in], bias [out])."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.normal_(self.weight, std=in_features ** -0.5)
        if bias:
            nn.init.zeros_(self.bias)

    def w(self, dtype):
        return self.cdt("w", self.weight, dtype)

    def b(self):
        return None if self.bias is None else self.f32("b", self.bias)

    def w8(self):
        """(e4m3 bytes [out, in], fp32 per-output-channel scales [out]) for the fp8 GEMM; quantised once per weight version."""
        pair = self.pk("w8", self.weight, lambda: ops.rows_fp8(self.weight.detach().contiguous()))
        return pair

    def tok(self, x: torch.Tensor, act: int = ops.ACT_NONE, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 token-side application (decoder tokens, hypernetwork / IoU MLPs): one wave per output for a handful of rows,
        the row-blocked kernel over the transposed weight from 64 rows up (many prompts)."""
        M = x.numel() // self.in_features
        if M >= 64 and self.in_features % 128 == 0:
            wt = self.pk("w32t", self.weight, lambda: self.weight.detach().float().t().contiguous())
            return ops.skinny_linear(x, wt, self.b(), act, res)
        return ops.small_linear(x, self.f32("w32", self.weight), self.b(), act, res)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = self.weight.dtype
        y = ops.gemm(ops.cast(x.reshape(-1, self.in_features).contiguous(), dt), self.w(dt), self.b())
        return y.reshape(*x.shape[:-1], self.out_features)


class LayerNorm(Packed):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.normalized_shape = (dim,)
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def wb(self):
        return self.f32("w", self.weight), self.f32("b", self.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w, b = self.wb()
        return ops.norm(x.contiguous(), w, b, self.eps, x.dtype)


class LayerNorm2d(Packed):
    """common.py:31-43.  Channel LayerNorm on NCHW == row LayerNorm on NHWC (biased variance, eps inside sqrt)."""

    def __init__(self, num_channels: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    def wb(self):
        return self.f32("w", self.weight), self.f32("b", self.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # NCHW in / out
        B, C, H, W = x.shape
        w, b = self.wb()
        nhwc = ops.transpose(x.float().contiguous().reshape(B, C, H * W), B, C, H * W)
        y = ops.norm(nhwc, w, b, self.eps, torch.float32)
        return ops.transpose(y, B, H * W, C).reshape(B, C, H, W).to(x.dtype)


class MLPBlock(Packed):
    """common.py:13-26: lin2(act(lin1(x)))."""

    def __init__(self, embedding_dim: int, mlp_dim: int, act: Type[nn.Module] = nn.GELU):
        super().__init__()
        self.lin1 = Linear(embedding_dim, mlp_dim)
        self.lin2 = Linear(mlp_dim, embedding_dim)
        self.act = act()
        self.act_code = ops.ACT_GELU if isinstance(self.act, nn.GELU) else ops.ACT_RELU

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = self.lin1.weight.dtype
        h = ops.gemm(ops.cast(x.reshape(-1, x.shape[-1]).contiguous(), dt), self.lin1.w(dt), self.lin1.b(), act=self.act_code)
        y = ops.gemm(h, self.lin2.w(dt), self.lin2.b())
        return y.reshape(x.shape)
