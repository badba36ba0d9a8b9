"""Checkpoint loading with the reference's conventions (train_joint_v2.py:1466-1555, app.py:335-341, build_sam.py:103-106).

The state_dict key layout of the HIP-backed modules is identical to the reference's, so its checkpoints load directly:
  * uLLSAM checkpoints (`final_all_e24.pt`): torch.save'd dict with the weights under "model" -> load_state_dict(strict=False)
  * SAM checkpoints (`sam_vit_b_01ec64.pth`): plain state_dict for Sam -> sam_model_registry[...](checkpoint=path)
  * InternVL2_5-2B `model.safetensors`: InternLM2 weights, re-prefixed with `language_model.` (train_joint_v2.py:1515-1548;
    the checkpoint's own `vision_model.*` / `mlp1.*` tensors belong to InternViT and are skipped, as in the reference)
No weight files ship with the reference or exist offline, so this path is exercised on synthetic
files with the real key layout (tests/test_host_cpu.py).
"""
from __future__ import annotations

import argparse
import os
import pathlib
from typing import Dict, Iterable, Tuple

import torch


def _report(model: torch.nn.Module, sd: Dict[str, torch.Tensor]) -> Tuple[list, list]:
    res = model.load_state_dict(sd, strict=False)
    return list(res.missing_keys), list(res.unexpected_keys)


def _torch_load(path: str, trusted: bool = False):
    """The reference saves {"model", "optimizer", "scheduler", "epoch", "step", "args": argparse.Namespace}
    (train_joint_v2.py:1254-1263): the Namespace (and any pathlib paths inside it) must be allow-listed for the weights-only
    unpickler.  `trusted=True` falls back to the full unpickler for files the caller vouches for."""
    allow = [argparse.Namespace, pathlib.PosixPath, pathlib.PurePosixPath, pathlib.Path]
    try:
        with torch.serialization.safe_globals(allow):
            return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        if not trusted:
            raise
        return torch.load(path, map_location="cpu", weights_only=False)


def load_ullsam_checkpoint(model: torch.nn.Module, path: str, trusted: bool = False) -> Tuple[list, list]:
    """`checkpoint["model"]` with strict=False (train_joint_v2.py:1472-1479); also accepts a bare state_dict."""
    ckpt = _torch_load(path, trusted)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict) else ckpt
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}  # DDP-wrapped saves
    return _report(model, sd)


def load_llm_safetensors(model: torch.nn.Module, path_or_dir: str, prefix: str = "language_model.") -> Tuple[list, list]:
    """InternLM2 weights from an InternVL-style safetensors file / directory (train_joint_v2.py:1515-1548)."""
    from safetensors.torch import load_file
    files: Iterable[str]
    if os.path.isdir(path_or_dir):
        files = sorted(os.path.join(path_or_dir, f) for f in os.listdir(path_or_dir) if f.endswith(".safetensors"))
    else:
        files = [path_or_dir]
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        for k, v in load_file(f).items():
            if k.startswith(prefix):
                sd[k] = v
            elif k.startswith(("model.", "output.")):
                sd[prefix + k] = v
    # the reference keeps only keys that exist with the same shape (train_joint_v2.py:1534-1545); the rest are skipped, not fatal
    have = model.state_dict()
    sd = {k: v for k, v in sd.items() if k in have and tuple(have[k].shape) == tuple(v.shape)}
    return _report(model, sd)


def prepack(model: torch.nn.Module, fp8_vit: bool = False) -> int:
    """Build the derived weight layouts once, right after loading, instead of inside the first forward ("into the
    build's weight layout ... pre-packing for MFMA tiles"): the compute-dtype copies and fp32 biases of every Linear, the interleaved
    [gate | up] rows of every InternLM2MLP (SwiGLU epilogue), and -- with fp8_vit -- the e4m3 bytes + per-channel scales of the ViT's qkv / lin1.
    The packs live in each module's PackCache keyed by the parameter version, so an optimizer step or load_state_dict invalidates them.
    Returns the number of packs built.  Needs the model on the GPU (the fp8 quantiser is a HIP kernel)."""
    from .modeling.common import Linear
    n = 0
    for name, m in model.named_modules():
        if isinstance(m, Linear):
            m.w(m.weight.dtype)
            m.b()
            n += 1
            if fp8_vit and m.weight.is_cuda and m.weight.dtype == torch.bfloat16 and (name.endswith("attn.qkv") or name.endswith("mlp.lin1")):
                m.w8()
                n += 1
        if hasattr(m, "w13") and hasattr(m, "w1"):
            m.w13(m.w1.weight.dtype)
            n += 1
    return n
