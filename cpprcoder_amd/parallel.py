"""Multi-GPU host side: one process per GPU, blocks sharded by rank, and the one real exchange
step of the path -- concatenating the per-rank compressed segments on every rank (allgatherv).

The reference has no multi-device code at all; blocks are independent, so coding
needs no collective.  RCCL has no native allgatherv; it is done here as one size all-gather, one fixed-size
all-gather of the segments padded to the largest, and a local compaction (or, optionally, one broadcast per
root).  The same code runs on gloo/CPU tensors, which is how the tests cover it.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_blocks(nblocks: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block range [first, last) of `rank`."""
    return nblocks * rank // world, nblocks * (rank + 1) // world


def allgatherv_segments(segment: torch.Tensor, seg_bytes: int, concat: torch.Tensor, group=None, staging: torch.Tensor | None = None,
                        method: str = "allgather"):
    """Concatenate every rank's first `seg_bytes` bytes of `segment` into `concat` on all ranks.

    RCCL has no allgatherv.  method="allgather" (default): one size all-gather, then ONE fixed-size
    all_gather_into_tensor of segments padded to the largest one (the collective RCCL is tuned for: per-block
    streams make the segments of equal-sized shards differ by well under 1 %), then a local compaction of the
    padded rows into `concat`.  method="broadcast": one broadcast per root straight into `concat` (no padding,
    no staging, but N separate collectives).

    Returns (bases, sizes, works): python lists of the per-rank base offset / size in `concat`, and async work
    handles to wait on before reading `concat` (empty for method="allgather", which completes on the current
    stream in order).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([seg_bytes], dtype=torch.int64, device=segment.device)
    every = torch.zeros(world, dtype=torch.int64, device=segment.device)
    dist.all_gather_into_tensor(every, mine, group=group)
    sizes = [int(x) for x in every.cpu()]
    bases = [0] * world
    for r in range(1, world):
        bases[r] = bases[r - 1] + sizes[r - 1]
    if bases[-1] + sizes[-1] > concat.numel():
        raise ValueError("concat buffer too small for the gathered segments")
    if method == "broadcast":
        concat[bases[rank]: bases[rank] + seg_bytes].copy_(segment[:seg_bytes])
        works = []
        for r in range(world):
            if sizes[r] == 0:
                continue
            view = concat[bases[r]: bases[r] + sizes[r]]
            src = dist.get_global_rank(group, r) if group is not None else r
            works.append(dist.broadcast(view, src=src, group=group, async_op=True))
        return bases, sizes, works
    row = (max(sizes) + 255) & ~255  # padded segment length, the same on every rank
    if row > segment.numel():
        raise ValueError("segment buffer shorter than the padded segment length")
    if staging is None or staging.numel() < world * row:
        staging = torch.empty(world * row, dtype=segment.dtype, device=segment.device)
    dist.all_gather_into_tensor(staging[: world * row], segment[:row], group=group)
    for r in range(world):
        if sizes[r]:
            concat[bases[r]: bases[r] + sizes[r]].copy_(staging[r * row: r * row + sizes[r]])
    return bases, sizes, []


def allgather_offsets(offsets: torch.Tensor, bases: list[int], group=None) -> torch.Tensor:
    """Global offset table from the per-rank ones (equal block counts per rank): rank r's local
    offsets shifted by its segment base; the final entry is the total size."""
    world = dist.get_world_size(group)
    per = offsets.numel() - 1
    table = torch.zeros(world * offsets.numel(), dtype=offsets.dtype, device=offsets.device)
    dist.all_gather_into_tensor(table, offsets.contiguous(), group=group)
    table = table.view(world, per + 1)
    shift = torch.tensor(bases, dtype=offsets.dtype, device=offsets.device).unsqueeze(1)
    shifted = table + shift
    return torch.cat([shifted[:, :per].reshape(-1), shifted[-1, per:]])
