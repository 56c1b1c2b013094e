"""The N > 1 host path on CPU: world_size-2 gloo, the same code bench.py runs over RCCL.

Each rank "encodes" its contiguous block shard with the oracle (there is no GPU here), then the
exchange step (allgatherv of the compressed segments + the global offset table) runs through
cpprcoder_amd.parallel exactly as on the GPU box, and every rank decodes the whole concatenated stream.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, ret):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle_lib
    from cpprcoder_amd import parallel, workloads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        block, nblocks = 4096, 49  # not divisible by the world size: the ranks bring different block counts
        whole = workloads.zipf(nblocks * block - 1000, 99)  # and the last block is short
        first, last = parallel.shard_blocks(nblocks, world, rank)
        shard = whole[first * block: last * block]
        chk = oracle_lib.oracle()
        slots, sizes = chk.encode_blocks(shard, block)
        payload, offsets = chk.compact(slots, sizes)
        assert len(sizes) == last - first
        seg = torch.zeros(len(payload) + 8192, dtype=torch.uint8)
        seg[: len(payload)] = torch.from_numpy(payload)
        concat = torch.zeros(world * (len(payload) + 8192), dtype=torch.uint8)
        table = torch.zeros(nblocks + 1, dtype=torch.int64)
        seg_base, block_base = parallel.allgatherv_p2p(seg, torch.from_numpy(offsets.astype(np.int64)), concat, table)
        assert block_base == [parallel.shard_blocks(nblocks, world, r)[0] for r in range(world)] + [nblocks]
        assert seg_base[rank + 1] - seg_base[rank] == len(payload)
        total = int(table[-1])
        assert total == seg_base[-1]
        # buffers that are too small are refused on every rank alike, before anything moves
        with pytest.raises(ValueError):
            parallel.allgatherv_p2p(seg, torch.from_numpy(offsets.astype(np.int64)), concat[: total - 1], table)
        with pytest.raises(ValueError):
            parallel.allgatherv_p2p(seg, torch.from_numpy(offsets.astype(np.int64)), concat, table[:nblocks])
        # ... also when only ONE rank's buffer is too small (the capacities travel with the sizes), or only one wants a table:
        # a rank that stopped alone would leave the others waiting in their receives
        with pytest.raises(ValueError):
            parallel.allgatherv_p2p(seg, torch.from_numpy(offsets.astype(np.int64)), concat[: total - 1] if rank == 1 else concat, table)
        with pytest.raises(ValueError):
            parallel.allgatherv_p2p(seg, torch.from_numpy(offsets.astype(np.int64)), concat, table if rank == 0 else None)
        # every rank can now decode the WHOLE buffer from the concatenated stream
        stream = concat[:total].numpy()
        tbl = table.numpy().astype(np.uint64)
        all_sizes = np.diff(tbl).astype(np.uint32)
        big = np.zeros((nblocks, chk.block_bound(block)), np.uint8)
        for b in range(nblocks):
            big[b, : all_sizes[b]] = stream[int(tbl[b]): int(tbl[b + 1])]
        back, ok = chk.decode_blocks(big, all_sizes, block, len(whole))
        assert ok and np.array_equal(back, whole)
        # and the concatenation equals a single-process encode of the whole buffer
        s1, z1 = chk.encode_blocks(whole, block)
        p1, o1 = chk.compact(s1, z1)
        assert np.array_equal(p1, stream) and np.array_equal(o1, tbl)
        ret[rank] = "ok"
    except Exception as e:  # pragma: no cover
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


def test_exchange_plan_is_the_librarys():
    # rcx_exchange_plan (include/rcx.h): pure host arithmetic, callable without a GPU
    from cpprcoder_amd import parallel, rcx
    assert parallel.exchange_plan([10, 0, 7], [2, 0, 1]) == ([0, 10, 10, 17], [0, 2, 2, 3])
    assert parallel.exchange_plan([5], [1]) == ([0, 5], [0, 1])
    with pytest.raises(rcx.RcxError):
        parallel.exchange_plan([1 << 63, 1 << 63], [1, 1])  # wraps


def test_shard_blocks():
    from cpprcoder_amd import parallel
    for nblocks in (1, 7, 16384, 131072):
        for world in (1, 2, 3, 8):
            cuts = [parallel.shard_blocks(nblocks, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nblocks
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))


@pytest.mark.parametrize("world", [2, 3])
def test_allgatherv_gloo(world):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.terminate()
            pytest.fail("gloo worker hung")
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)
