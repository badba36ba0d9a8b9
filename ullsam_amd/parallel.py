"""Image-level data parallelism for inference: one process per GPU, weights replicated, images sharded, ONE exchange step
(an RCCL all-gather over xGMI of the masks / token ids) at the end.  SURVEY.md section 8(e): the reference has nothing
comparable (app.py:39 hard-codes one GPU, batch 1), so this is the build's own, minimal, collective.

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous split of n_items over ranks; the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, counts: Optional[List[int]] = None, group=None) -> torch.Tensor:
    """Concatenate every rank's rows (dim 0) in rank order.

    Equal shards use a single all_gather_into_tensor (one flat all-gather: each rank's slice goes to all 7 peers over its
    direct xGMI links).  Ragged shards (counts differ) are padded to the largest shard for the collective and trimmed after."""
    rank, ws = world()
    if ws == 1:
        return local
    local = local.contiguous()
    if counts is None:
        c = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        allc = [torch.zeros_like(c) for _ in range(ws)]
        dist.all_gather(allc, c, group=group)
        counts = [int(t.item()) for t in allc]
    mx = max(counts)
    if all(n == mx for n in counts):
        out = torch.empty((ws * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((ws * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(ws)], 0)


def gather_mask_results(low_res_logits: torch.Tensor, masks_u8: torch.Tensor, token_ids: Optional[torch.Tensor] = None,
                        counts: Optional[List[int]] = None):
    """The final exchange of the path: low-res logits [b,1,256,256] fp32 (256 KiB/img), thresholded masks [b,1,1024,1024] u8
    (1 MiB/img) and, for caption runs, greedy token ids padded to a common length."""
    low = all_gather_rows(low_res_logits, counts)
    mk = all_gather_rows(masks_u8, counts)
    tok = all_gather_rows(token_ids, counts) if token_ids is not None else None
    return low, mk, tok
