"""Multi-GPU host side: one process per GPU, blocks sharded by rank, and the one real exchange step of the path --
putting the per-rank compressed segments and their block tables together on every rank ("allgatherv").

The reference has no multi-device code at all; blocks are independent, so coding needs no
collective.  The exchange itself lives in the C ABI (include/rcx.h: rcx_comm_*, rcx_allgatherv_segments -- RCCL
point-to-point sends and receives straight into place); `Comm` below is its binding.  `allgatherv_p2p` is the same
exchange written over torch.distributed point-to-point calls: it runs on gloo/CPU tensors, which is how the tests
cover the offset arithmetic without a GPU; both use the same plan function of the library (rcx_exchange_plan).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import rcx


def shard_blocks(nblocks: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block range [first, last) of `rank`; ranks may get unequal counts."""
    return nblocks * rank // world, nblocks * (rank + 1) // world


def exchange_plan(seg_bytes, nblocks) -> tuple[list[int], list[int]]:
    """Where every rank's bytes and blocks start in the concatenation (rcx_exchange_plan; pure, no GPU)."""
    n = len(seg_bytes)
    seg = (C.c_uint64 * n)(*[int(x) for x in seg_bytes])
    blk = (C.c_uint64 * n)(*[int(x) for x in nblocks])
    seg_base, block_base = (C.c_uint64 * (n + 1))(), (C.c_uint64 * (n + 1))()
    st = rcx.lib().rcx_exchange_plan(seg, blk, n, seg_base, block_base)
    if st != rcx.OK:
        raise rcx.RcxError(st, "rcx_exchange_plan")
    return list(seg_base), list(block_base)


def allgatherv_p2p(segment: torch.Tensor, offsets: torch.Tensor, concat: torch.Tensor, table: torch.Tensor | None, group=None):
    """The exchange of rcx_allgatherv_segments over torch.distributed point-to-point calls (any backend).

    segment: this rank's compacted streams; offsets: its nblocks+1 table (offsets[-1] = segment bytes); concat receives
    the segments back to back; table (optional) the table of the concatenation.  Returns (seg_base, block_base).
    Every rank raises together if a buffer is too small (the plan is the same everywhere)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nblocks = offsets.numel() - 1
    # {segment bytes, blocks, room in concat, room in the table, table wanted}: the capacities travel with the sizes, so every
    # rank judges by the smallest room ANY rank has and they all refuse together (csrc/rcx_comm.hip does the same)
    words = [int(offsets[-1]), nblocks, concat.numel(), table.numel() if table is not None else 0, 1 if table is not None else 0]
    mine = torch.tensor(words, dtype=torch.int64, device=offsets.device)
    every = torch.zeros(5 * world, dtype=torch.int64, device=offsets.device)
    dist.all_gather_into_tensor(every, mine, group=group)
    every = every.cpu().view(world, 5)
    seg, blk = [int(x) for x in every[:, 0]], [int(x) for x in every[:, 1]]
    seg_base, block_base = exchange_plan(seg, blk)
    wanted = int(every[:, 4].sum())
    if wanted not in (0, world):
        raise ValueError("some ranks pass a table and some do not")
    if seg_base[-1] > int(every[:, 2].min()) or (table is not None and block_base[-1] + 1 > int(every[:, 3].min())):
        raise ValueError("concat / table too small for the gathered segments on some rank")
    concat[seg_base[rank]: seg_base[rank] + seg[rank]].copy_(segment[: seg[rank]])
    if table is not None:
        table[block_base[rank]: block_base[rank] + nblocks].copy_(offsets[:nblocks] + seg_base[rank])
        if rank == world - 1:
            table[block_base[-1]] = seg_base[-1]
    ops = []
    glob = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    for step in range(1, world):
        to, frm = (rank + step) % world, (rank - step) % world
        if seg[rank]:
            ops.append(dist.P2POp(dist.isend, segment[: seg[rank]], glob(to), group))
        if seg[frm]:
            ops.append(dist.P2POp(dist.irecv, concat[seg_base[frm]: seg_base[frm] + seg[frm]], glob(frm), group))
        if table is not None:
            mine_n, theirs_n = blk[rank] + (rank == world - 1), blk[frm] + (frm == world - 1)
            if mine_n:
                ops.append(dist.P2POp(dist.isend, table[block_base[rank]: block_base[rank] + mine_n], glob(to), group))
            if theirs_n:
                ops.append(dist.P2POp(dist.irecv, table[block_base[frm]: block_base[frm] + theirs_n], glob(frm), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return seg_base, block_base


class Comm:
    """One rcx_comm: this process's GPU in the node's RCCL communicator (include/rcx.h)."""

    def __init__(self, device: int, unique_id: bytes, nranks: int, rank: int):
        self._h = C.c_void_p()
        buf = (C.c_char * rcx.COMM_ID_BYTES).from_buffer_copy(unique_id)
        st = rcx.lib().rcx_comm_create(device, buf, nranks, rank, C.byref(self._h))
        if st != rcx.OK:
            raise rcx.RcxError(st, "rcx_comm_create")
        self.nranks, self.rank, self.device = nranks, rank, device

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * rcx.COMM_ID_BYTES)()
        st = rcx.lib().rcx_comm_unique_id(buf)
        if st != rcx.OK:
            raise rcx.RcxError(st, "rcx_comm_unique_id")
        return bytes(buf.raw)

    @classmethod
    def from_process_group(cls, device: int, group=None) -> "Comm":
        """Rank 0 makes the id and torch.distributed carries it to the others (the "out of band" channel)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        return cls(device, ident[0], world, rank)

    def close(self) -> None:
        if getattr(self, "_h", None):
            rcx.lib().rcx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allgatherv(self, segment, offsets, concat, table=None, stream=None):
        """rcx_allgatherv_segments: segment/offsets as the encoder wrote them -> concat (+ table) on every rank.
        Returns (seg_base, block_base) as python lists of nranks+1."""
        nblocks = offsets.numel() - 1
        seg_base, block_base = (C.c_uint64 * (self.nranks + 1))(), (C.c_uint64 * (self.nranks + 1))()
        st = rcx.lib().rcx_allgatherv_segments(self._h, segment.data_ptr(), offsets.data_ptr(), nblocks, concat.data_ptr(), concat.numel(),
                                               table.data_ptr() if table is not None else None, table.numel() if table is not None else 0,
                                               seg_base, block_base, rcx.Context._stream_handle(stream))
        if st != rcx.OK:
            raise rcx.RcxError(st, "rcx_allgatherv_segments")
        return list(seg_base), list(block_base)
