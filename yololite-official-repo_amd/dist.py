"""Multi-GPU: one process per GPU, images sharded by index, ONE collective at the end.

The reference is single-process / single-device (no collective anywhere).  Images are
independent (BatchNorm in eval mode, no cross-image op in the hot path), so a batch shards as
contiguous slices with replicated weights and NO data-path collective; the only exchange is an
all-gather of the packed per-image results (dets [b,max_out,6] fp32 + counts [b] int32; 7.2 KB per
image at max_out=300) over RCCL/xGMI (`backend="nccl"` is RCCL on ROCm).  Payload is latency-bound.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of `total` images for `rank` (first total%world ranks get one more)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_dets(dets: torch.Tensor, counts: torch.Tensor, total: int, group=None,
                   force: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """dets [b_local,max_out,6], counts [b_local] of this rank's shard -> ([total,max_out,6], [total]) in image
    order on every rank.  Shards may differ by one image; they are padded to the largest shard for a single
    all_gather_into_tensor of a packed buffer (dets rows and the count travel together)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return dets, counts
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    max_out = dets.shape[1]
    per = -(-total // world)
    row = max_out * 6 + 1
    pack = torch.zeros((per, row), device=dets.device, dtype=torch.float32)
    b = dets.shape[0]
    pack[:b, :max_out * 6] = dets.reshape(b, -1)
    pack[:b, max_out * 6] = counts.to(torch.float32)           # exact for counts < 2^24
    out = torch.empty((world * per, row), device=dets.device, dtype=torch.float32)
    dist.all_gather_into_tensor(out, pack, group=group)
    out = out.view(world, per, row)
    pieces_d, pieces_c = [], []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        n = hi - lo
        pieces_d.append(out[r, :n, :max_out * 6].reshape(n, max_out, 6))
        pieces_c.append(out[r, :n, max_out * 6].to(torch.int32))
    return torch.cat(pieces_d, 0), torch.cat(pieces_c, 0)


def sharded_predict(predict_fn, x_global_cpu_or_dev: torch.Tensor, device, group=None):
    """Run `predict_fn(x_shard) -> (dets, counts)` on this rank's slice of the batch and all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    total = x_global_cpu_or_dev.shape[0]
    lo, hi = shard_range(total, rank, world)
    dets, counts = predict_fn(x_global_cpu_or_dev[lo:hi].to(device))
    return allgather_dets(dets, counts, total, group)


class DetGatherer:
    """Zero-copy, pipelined exchange for equal shards (the serving / benchmark case: b_local images on every
    rank).  The per-rank result lives in ONE flat buffer [dets (b*max_out*6 floats) | counts (b int32)] that
    yl_predict writes in place (`.dets`, `.counts` are views of the CURRENT slot); `gather()` starts a single
    asynchronous all_gather_into_tensor of that buffer into a preallocated [world, row] tensor and flips to the
    other slot, so the collective of step i overlaps the forward pass of step i+1 (no pack / unpack kernels, no
    host-blocking wait: on ROCm 7 a synchronous all_gather issued behind a busy compute stream costs ~0.7 ms,
    the asynchronous one ~0.03 ms -- measured in round 1 with a throw-away probe; tools/rccl_queue_probe.py is the
    kept tool for RCCL next to the executor's streams).  `gather()` returns the PREVIOUS step's result views
    (None on the first call); `flush()` waits for everything in flight and returns the last one.
    Results: dets [world, b, max_out, 6], counts [world, b]; image i of the global batch is (i // b, i % b)."""

    def __init__(self, b_local: int, max_out: int, device, group=None, slots: int = 2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.b, self.max_out = int(b_local), int(max_out)
        nd = self.b * self.max_out * 6
        self._nd = nd
        self._loc = [torch.zeros((nd + self.b,), device=device, dtype=torch.float32) for _ in range(slots)]
        self._out = [torch.zeros((self.world, nd + self.b), device=device, dtype=torch.float32) for _ in range(slots)]
        self._work = [None] * slots
        self._k = 0
        self._last = None

    @property
    def dets(self) -> torch.Tensor:
        return self._loc[self._k][:self._nd].view(self.b, self.max_out, 6)

    @property
    def counts(self) -> torch.Tensor:
        return self._loc[self._k][self._nd:].view(torch.int32)

    def _views(self, k):
        return (self._out[k][:, :self._nd].view(self.world, self.b, self.max_out, 6),
                self._out[k][:, self._nd:].view(torch.int32))

    def gather(self):
        k = self._k
        if dist.is_initialized():
            self._work[k] = dist.all_gather_into_tensor(self._out[k].view(-1), self._loc[k], group=self.group,
                                                        async_op=True)
        else:
            self._out[k][0].copy_(self._loc[k])
        prev, self._last = self._last, k
        self._k = (k + 1) % len(self._loc)
        nxt = self._k                      # the slot yl_predict writes next must not be in flight any more
        if self._work[nxt] is not None:
            self._work[nxt].wait()
            self._work[nxt] = None
        return None if prev is None else self._views(prev)

    def flush(self):
        for i, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[i] = None
        return None if self._last is None else self._views(self._last)
