"""Image-level data parallelism for inference: one process per GPU, weights replicated, images sharded, ONE exchange step
(an RCCL all-gather over xGMI of the masks / token ids) at the end.  SURVEY.md section 8(e): the reference has nothing
comparable (app.py:39 hard-codes one GPU, batch 1; its only launcher is scripts/train_all_joint_v2.sh:1, torchrun for
training), so this is the build's own, minimal, collective.

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.

The exchange is ONE collective per step: logits (fp32), masks (u8) and token ids (int64) of a rank's images are packed
per image into one byte record, the records of all ranks are gathered with a single all_gather_into_tensor (flat
all-gather: every rank's slice travels over its 7 direct xGMI links; at 4 images per rank it is 5 MiB per rank,
latency-bound), and unpacked on arrival.  `gather_mask_results_async` returns a handle so the caller can overlap the
exchange of step k with the compute of step k+1.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world(group=None) -> Tuple[int, int]:
    """(rank, size) in `group` (default: the default process group); (0, 1) without torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous split of n_items over ranks; the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def exchange_counts(n_local: int, device, group=None) -> List[int]:
    """Rows held by every rank (one tiny collective; callers that know the split pass `counts` and skip it)."""
    _, ws = world(group)
    if ws == 1:
        return [n_local]
    c = torch.tensor([n_local], dtype=torch.int64, device=device)
    allc = torch.empty((ws,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, c, group=group)
    return [int(v) for v in allc.tolist()]


class _Pending:
    """Handle of an in-flight packed all-gather: wait() -> the gathered tensors, rows in rank order."""

    def __init__(self, work, out, specs, counts, mx, rec=None):
        # `rec` (the local send buffer) is held until wait(): the collective reads it asynchronously
        self.work, self.out, self.specs, self.counts, self.mx, self.rec = work, out, specs, counts, mx, rec

    def wait(self):
        if self.work is not None:
            self.work.wait()  # the current stream waits for the collective (nccl) / the call blocks (gloo)
        self.rec = None
        ws = len(self.counts)
        rec = self.out.reshape(ws, self.mx, -1)
        if any(n != self.mx for n in self.counts):
            rec = torch.cat([rec[r, : self.counts[r]] for r in range(ws)], 0)
        else:
            rec = rec.reshape(ws * self.mx, -1)
        res, off = [], 0
        for shape, dtype, nbytes in self.specs:
            if shape is None:
                res.append(None)
                continue
            res.append(rec[:, off:off + nbytes].contiguous().view(dtype).reshape((rec.shape[0],) + shape))
            off += nbytes
        return res


def packed_all_gather(tensors: Sequence[Optional[torch.Tensor]], counts: Optional[List[int]] = None, group=None,
                      async_op: bool = False):
    """Gather several per-image tensors (same dim 0 on a rank) with ONE all_gather_into_tensor.  Ragged shards are padded to the
    largest shard for the collective and trimmed after.  Returns a handle (async_op) or the list of gathered tensors."""
    rank, ws = world(group)
    live = [t for t in tensors if t is not None]
    n = live[0].shape[0]
    dev = live[0].device
    if counts is None:
        counts = exchange_counts(n, dev, group)
    assert counts[rank] == n, (counts, rank, n)
    mx = max(counts)
    specs, parts = [], []
    for t in tensors:
        if t is None:
            specs.append((None, None, 0))
            continue
        assert t.shape[0] == n
        # bytes per row from the shape, not from reshape(n, -1): a rank may hold zero rows (n_items < world size)
        row_bytes = math.prod(t.shape[1:]) * t.element_size()
        b = t.contiguous().view(torch.uint8).reshape(n, row_bytes)
        specs.append((tuple(t.shape[1:]), t.dtype, row_bytes))
        parts.append(b)
    width = sum(s[2] for s in specs)
    # one send buffer of the padded shard size, filled in place (no zero-fill + copy of the whole record for ragged shards:
    # the pad rows are trimmed on arrival and never read)
    rec = torch.empty((mx, width), dtype=torch.uint8, device=dev)
    off = 0
    for b in parts:
        rec[:n, off:off + b.shape[1]] = b
        off += b.shape[1]
    if not (dist.is_available() and dist.is_initialized()):
        p = _Pending(None, rec, specs, counts, mx)
        return p if async_op else p.wait()
    out = torch.empty((ws * mx, width), dtype=torch.uint8, device=dev)
    work = dist.all_gather_into_tensor(out, rec, group=group, async_op=async_op)
    p = _Pending(work if async_op else None, out, specs, counts, mx, rec)
    return p if async_op else p.wait()


def all_gather_rows(local: torch.Tensor, counts: Optional[List[int]] = None, group=None) -> torch.Tensor:
    """Concatenate every rank's rows (dim 0) in rank order (single collective when `counts` is given)."""
    return packed_all_gather([local], counts, group)[0]


def gather_mask_results(low_res_logits: torch.Tensor, masks_u8: torch.Tensor, token_ids: Optional[torch.Tensor] = None,
                        counts: Optional[List[int]] = None, group=None):
    """The final exchange of the path: low-res logits [b,1,256,256] fp32 (256 KiB/img), thresholded masks [b,1,1024,1024] u8
    (1 MiB/img) and, for caption runs, greedy token ids padded to a common length -- one collective for all three."""
    low, mk, tok = packed_all_gather([low_res_logits, masks_u8, token_ids], counts, group)
    return low, mk, tok


def gather_mask_results_async(low_res_logits: torch.Tensor, masks_u8: torch.Tensor, token_ids: Optional[torch.Tensor] = None,
                              counts: Optional[List[int]] = None, group=None) -> _Pending:
    """Same exchange, returned as a handle: `.wait()` gives (logits, masks, token ids).  Lets step k+1's compute overlap it."""
    return packed_all_gather([low_res_logits, masks_u8, token_ids], counts, group, async_op=True)


def gather_sharded_lists(local: list, n_items: int, group=None) -> list:
    """Units of work (tiles of the automatic mask generator, BASELINE configs[4]) are dealt to the ranks by `shard_range`; every rank passes the
    results of ITS units (a list with one entry per local unit, each entry any picklable object -- e.g. a tile's ragged list of mask records) and
    gets back the results of all `n_items` units in unit order.  One collective (all_gather_object: sizes, then the pickled payloads as byte
    tensors over RCCL / gloo); a rank that holds zero units still takes part."""
    rank, ws = world(group)
    a, b = shard_range(n_items, rank, ws)
    assert len(local) == b - a, (len(local), a, b)
    if ws == 1:
        return list(local)
    parts = [None] * ws
    dist.all_gather_object(parts, list(local), group=group)
    out = []
    for r in range(ws):
        ra, rb = shard_range(n_items, r, ws)
        assert len(parts[r]) == rb - ra, (r, len(parts[r]), ra, rb)
        out.extend(parts[r])
    return out
