"""rcx_comm_* / rcx_allgatherv_segments (include/rcx.h) on RCCL with the one rank a 1-GPU box has: communicator
set-up, the size all-gather, the in-place copy of the own segment and the shifted block table.  The N > 1 offset
arithmetic (rcx_exchange_plan + the point-to-point pattern) is covered on CPU by tests/test_parallel_gloo.py; the
reference has no multi-device code to compare with."""
import numpy as np
import pytest

from cpprcoder_amd import workloads

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_one_rank_allgatherv_over_rccl(oracle):
    from cpprcoder_amd import parallel, rcx
    assert torch.cuda.is_available()
    ctx = rcx.Context(0)
    comm = parallel.Comm(0, parallel.Comm.unique_id(), 1, 0)
    lib = rcx.lib()
    assert lib.rcx_comm_rank(comm._h) == 0 and lib.rcx_comm_size(comm._h) == 1
    n, block = 65536 * 5 + 321, 65536
    data = workloads.zipf(n, 5)
    nblocks = rcx.block_count(n, block)
    src = torch.from_numpy(data).cuda()
    seg = torch.zeros(rcx.encode_bound(n, block), dtype=torch.uint8, device="cuda")
    offs = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(src, block, seg, offs)
    ctx.sync_status()
    total = int(offs[-1])
    concat = torch.zeros(total + 100, dtype=torch.uint8, device="cuda")
    table = torch.full((nblocks + 1,), -1, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        seg_base, block_base = comm.allgatherv(seg, offs, concat, table)
    side.synchronize()
    assert seg_base == [0, total] and block_base == [0, nblocks]
    assert torch.equal(concat[:total], seg[:total]) and torch.equal(table, offs)
    # the concatenation decodes (it is the stream of the whole buffer) and equals the oracle's
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks_device(concat, total, table, n, block, out)
    ctx.sync_status()
    assert torch.equal(out, src)
    slots, sizes = oracle.encode_blocks(data, block, threads=4)
    payload, offsets = oracle.compact(slots, sizes)
    assert np.array_equal(concat[:total].cpu().numpy(), payload) and np.array_equal(table.cpu().numpy().astype(np.uint64), offsets)
    # too small: refused before anything moves
    with pytest.raises(rcx.RcxError) as e:
        comm.allgatherv(seg, offs, concat[: total - 1], table)
    assert e.value.status == rcx.E_CAPACITY
    with pytest.raises(rcx.RcxError):
        comm.allgatherv(seg, offs, concat, table[:nblocks])
    comm.close()
    ctx.close()
