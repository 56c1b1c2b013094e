"""The host-buffer entry points (rcx_encode_blocks / rcx_decode_blocks / rcx_bwt_encode / rcx_bwt_decode) made in chunks
(csrc/rcx_host.hpp): the reference is driven with host memory on both sides (test/main.cpp:321-350), so this is the
path a drop-in caller takes.  Every block's stream must be what the device-pointer call -- and the oracle -- produce,
whichever way the pieces cross the link, and the per-chunk tables must add up to the one table of the whole buffer.
"""
import numpy as np
import pytest

import bwt_cases
import oracle_lib
from cpprcoder_amd import workloads

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

BLOCK = 4096                      # small blocks: chunks are about 4096 blocks = 16 MiB, so a modest buffer spans many
N = 5 * (16 << 20) + (9 << 20) + 1234  # 6 chunks and a ragged last block (1234 bytes)


def chunk_blocks(block, decode, nblocks):
    """csrc/rcx_host.hpp host_chunk_blocks(): how many blocks the host-buffer calls put into a chunk."""
    cb = max(4096 if decode else 2048, -(-(16 << 20) // block))
    while -(-nblocks // cb) > 2048:
        cb *= 2
    chunks = -(-nblocks // cb)
    if chunks > 1:
        cb = -(-nblocks // chunks)
    return (cb + 63) & ~63


@pytest.fixture(scope="module")
def data():
    return np.concatenate([workloads.by_name("canterbury", 40 << 20, 1), workloads.by_name("zipf", N - (40 << 20), 7)])


def device_encode(ctx, data, block, coder):
    from cpprcoder_amd import rcx
    src = torch.from_numpy(data).cuda()
    nblocks = rcx.block_count(len(data), block)
    dst = torch.zeros(rcx.encode_bound(len(data), block, coder), dtype=torch.uint8, device="cuda")
    offs = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(src, block, dst, offs, coder=coder)
    ctx.sync_status()
    offsets = offs.cpu().numpy().astype(np.uint64)
    return dst[: int(offsets[-1])].cpu().numpy(), offsets


def host_round_trip(ctx, data, block, coder):
    from cpprcoder_amd import rcx
    n = len(data)
    dst = np.full(rcx.encode_bound(n, block, coder), 0xA5, dtype=np.uint8)
    offsets = np.zeros(rcx.block_count(n, block) + 1, dtype=np.uint64)
    size = ctx.encode_blocks_into(data, block, dst, offsets, coder)
    out = np.full(n + 64, 0x5A, dtype=np.uint8)
    got = ctx.decode_blocks_into(dst, size, offsets, block, out[:n], coder)
    assert got == n and bool((out[n:] == 0x5A).all())
    return dst, size, offsets, out[:n]


@pytest.mark.parametrize("mode", ["direct", "staged", "register"])
def test_chunked_host_calls_equal_the_device_call(monkeypatch, data, oracle, mode):
    from cpprcoder_amd import rcx
    monkeypatch.setenv("RCX_HOST_MODE", mode)
    monkeypatch.setenv("RCX_HOST_MOVERS", "3")
    monkeypatch.setenv("RCX_HOST_PIECE_MIB", "3")  # pieces that do not divide a chunk
    ctx = rcx.Context(0)
    try:
        dst, size, offsets, back = host_round_trip(ctx, data, BLOCK, rcx.CODER_ADAPTIVE)
        assert np.array_equal(back, data)
        payload, dev_offsets = device_encode(ctx, data, BLOCK, rcx.CODER_ADAPTIVE)
        assert size == len(payload) and np.array_equal(offsets, dev_offsets)
        assert np.array_equal(dst[:size], payload)
        assert bool((dst[size:] == 0xA5).all()), "wrote past the size it reported"
        # and against the oracle, block by block (the first chunk boundary, the ragged end, a sample in between)
        nblocks = len(offsets) - 1
        cb = chunk_blocks(BLOCK, False, nblocks)
        assert nblocks // cb >= 5, "the buffer should span at least 5 whole chunks"
        for b in [0, 1, cb - 1, cb, cb + 1, 2 * cb - 1, 2 * cb, nblocks // 2, nblocks - 2, nblocks - 1]:
            piece = data[b * BLOCK: (b + 1) * BLOCK]
            slots, sizes = oracle.encode_blocks(piece, BLOCK)
            assert np.array_equal(dst[int(offsets[b]): int(offsets[b + 1])], slots[0, : int(sizes[0])]), f"block {b}"
    finally:
        ctx.close()


@pytest.mark.parametrize("coder", [1, 2, 3])
def test_chunked_host_calls_sibling_coders(data, coder):
    from cpprcoder_amd import rcx
    ctx = rcx.Context(0)
    try:
        part = data[: 3 * (16 << 20) + 777]
        dst, size, offsets, back = host_round_trip(ctx, part, BLOCK, coder)
        assert np.array_equal(back, part)
        payload, dev_offsets = device_encode(ctx, part, BLOCK, coder)
        assert np.array_equal(offsets, dev_offsets) and np.array_equal(dst[:size], payload)
    finally:
        ctx.close()


def test_chunked_host_calls_64k_blocks(data, oracle):
    """BASELINE.json's block size: 64 KiB blocks, chunks of 1024 (encode) and 2048 (decode) blocks."""
    from cpprcoder_amd import rcx
    ctx = rcx.Context(0)
    try:
        big = np.concatenate([data, data[::-1], data[: (70 << 20) + 4321]])  # ~ 267 MiB: 5 encode chunks, 3 decode chunks
        dst, size, offsets, back = host_round_trip(ctx, big, 65536, rcx.CODER_ADAPTIVE)
        assert np.array_equal(back, big)
        slots, sizes = oracle.encode_blocks(big, 65536, threads=16)
        assert np.array_equal(np.diff(offsets.astype(np.int64)), sizes.astype(np.int64))
        for b in range(0, len(sizes), 37):
            assert np.array_equal(dst[int(offsets[b]): int(offsets[b + 1])], slots[b, : int(sizes[b])]), f"block {b}"
        b = len(sizes) - 1
        assert np.array_equal(dst[int(offsets[b]): int(offsets[b + 1])], slots[b, : int(sizes[b])]), "the ragged last block"
    finally:
        ctx.close()


def test_chunked_host_calls_report_errors(data):
    from cpprcoder_amd import rcx
    import ctypes as C
    ctx = rcx.Context(0)
    try:
        n = len(data)
        nblocks = rcx.block_count(n, BLOCK)
        dst = np.zeros(rcx.encode_bound(n, BLOCK), dtype=np.uint8)
        offsets = np.zeros(nblocks + 1, dtype=np.uint64)
        size = ctx.encode_blocks_into(data, BLOCK, dst, offsets)
        # a destination that is too small: refused, and the size needed is reported
        small = np.full(size - 100 + 64, 0xA5, dtype=np.uint8)
        need = C.c_uint64()
        st = rcx.lib().rcx_encode_blocks(ctx._h, 0, data.ctypes.data, n, BLOCK, small.ctypes.data, size - 100, C.byref(need), None)
        assert st == rcx.E_CAPACITY and need.value == size
        assert bool((small[size - 100:] == 0xA5).all())
        # a table that runs backwards, or past the payload
        out = np.zeros(n, dtype=np.uint8)
        bad = offsets.copy()
        bad[4096] = bad[8192] + 1
        with pytest.raises(rcx.RcxError) as e:
            ctx.decode_blocks_into(dst, size, bad, BLOCK, out)
        assert e.value.status == rcx.E_CORRUPT
        with pytest.raises(rcx.RcxError) as e:
            ctx.decode_blocks_into(dst, size - 5, offsets, BLOCK, out)
        assert e.value.status == rcx.E_CORRUPT
        # a damaged block in the middle chunk: reported, the call returns, and the context still works
        hurt = dst[:size].copy()
        hurt[int(offsets[6000]): int(offsets[6000]) + 4] = 0xFF  # the block's declared size
        with pytest.raises(rcx.RcxError) as e:
            ctx.decode_blocks_into(hurt, size, offsets, BLOCK, out)
        assert e.value.status == rcx.E_CORRUPT
        assert ctx.decode_blocks_into(dst, size, offsets, BLOCK, out) == n and np.array_equal(out, data)
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", ["direct", "staged"])
def test_chunked_block_sort(monkeypatch, data, oracle, mode):
    from cpprcoder_amd import rcx
    monkeypatch.setenv("RCX_HOST_MODE", mode)
    ctx = rcx.Context(0)
    try:
        # 3 chunks of 1024 blocks + a part, a periodic block in the second chunk (its row index is the reference sort's)
        # and a tail shorter than a block
        body = data[: 3 * (32 << 20) + 5 * 32768].copy()
        body[1500 * 32768: 1501 * 32768] = bwt_cases.periodic(16, 14)
        src = np.concatenate([body, data[:999]])
        enc = np.full(rcx.bwt_encode_bound(len(src)) + 64, 0xA5, dtype=np.uint8)
        m = ctx.bwt_into(True, src, enc[:-64])
        assert m == rcx.bwt_encode_bound(len(src)) and bool((enc[m:] == 0xA5).all())
        assert ctx.bwt_last_ties() == 1
        back = np.zeros(len(src), dtype=np.uint8)
        assert ctx.bwt_into(False, enc[:m], back) == len(src) and np.array_equal(back, src)
        # against the oracle: the chunk boundaries, the periodic block, the tail
        for b in [0, 1023, 1024, 1500, 2047, 2048, 3072, 3076]:
            want = oracle.bwt_encode(src[b * 32768: (b + 1) * 32768])
            assert np.array_equal(enc[b * 32770: (b + 1) * 32770], want), f"block {b}"
        assert np.array_equal(enc[3077 * 32770: m], src[3077 * 32768:])
    finally:
        ctx.close()


def test_an_error_left_behind_by_another_hip_call_is_not_reported(data):
    """Since HIP 7 an error code returned by any runtime call stays in the calling thread's "last error" until it is read.  The
    library checks its kernel launches by reading it, so it clears it on entry: an error some earlier call of the thread left
    behind (here: a hipFree of nonsense) must not turn up as RCX_E_HIP of the next encode."""
    import ctypes as C
    from cpprcoder_amd import rcx
    rcx.lib()
    path = next((line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line), None)
    assert path, "the HIP runtime should be loaded by now"
    hip = C.CDLL(path)
    hip.hipFree.argtypes, hip.hipFree.restype = [C.c_void_p], C.c_int
    assert hip.hipFree(C.c_void_p(0x1234)) != 0  # an error that nobody reads
    ctx = rcx.Context(0)
    try:
        part = data[: 3 * (16 << 20) + 99]
        dst, size, offsets, back = host_round_trip(ctx, part, BLOCK, rcx.CODER_ADAPTIVE)
        assert np.array_equal(back, part)
        # the device-pointer calls likewise (the tensors are made first: torch reads the thread's last error too, and would
        # report the one provoked here as its own)
        src = torch.from_numpy(part).cuda()
        out = torch.zeros(rcx.encode_bound(len(part), BLOCK), dtype=torch.uint8, device="cuda")
        offs = torch.zeros(rcx.block_count(len(part), BLOCK) + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        assert hip.hipFree(C.c_void_p(0x1234)) != 0
        ctx.encode_blocks_device(src, BLOCK, out, offs)
        ctx.sync_status()
        hip.hipGetLastError.restype = C.c_int
        hip.hipGetLastError()  # (nothing of ours should be there; whatever is, is not left for torch to find)
        total = int(offs[-1])
        assert total == size and np.array_equal(out[:total].cpu().numpy(), dst[:size])
    finally:
        ctx.close()
