"""Host-side weight re-layouts for the HIP kernels (done once per weight version; torch is plumbing here)."""
from __future__ import annotations

import torch


def pack_w13(w1: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    """[2I, K] with alternating 64-row blocks (w1 block t, w3 block t): a 128-column GEMM tile then holds the gate and
    up projections of the same 64 outputs, so silu(w1 x) * w3 x (modeling_internlm2.py:261-264) is fused in the epilogue."""
    I, K = w1.shape
    assert w3.shape == (I, K) and I % 64 == 0
    return torch.stack([w1.reshape(I // 64, 64, K), w3.reshape(I // 64, 64, K)], dim=1).reshape(2 * I, K).contiguous()


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [Cout, Cin, 3, 3] -> [Cout, (ky, kx, Cin)] matching ullsam_im2col3x3's column order."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_convT_k2s2(w: torch.Tensor, b: torch.Tensor):
    """ConvTranspose2d(k=2, s=2) weight [Cin, Cout, 2, 2] -> GEMM weight [(ky, kx, Cout), Cin] and bias tiled x4
    (stride == kernel: each output pixel gets exactly one tap, mask_decoder.py:53-59)."""
    cin, cout = w.shape[0], w.shape[1]
    return w.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous(), b.repeat(4).contiguous()
