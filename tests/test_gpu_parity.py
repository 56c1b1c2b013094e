"""GPU parity: the HIP path (through the C ABI) against the oracle and the golden vectors.

Bit-exact bar: every block's stream must equal what the reference emits for that block
(cpprcoder.h:678-802), and decode(encode(x)) == x (the reference harness's own check,
test/main.cpp:357-361).  Nothing here reads /root/reference.
"""
import hashlib

import numpy as np
import pytest

import oracle_lib
from cpprcoder_amd import workloads

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    from cpprcoder_amd import rcx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = rcx.Context(0)
    yield c
    c.close()


def gpu_encode(ctx, data, block, src_offset=0, coder=0):
    """-> (payload np.uint8, offsets np.uint64) through the device-pointer entry points."""
    from cpprcoder_amd import rcx
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = len(data)
    buf = torch.zeros(n + src_offset + 16, dtype=torch.uint8, device="cuda")
    buf[src_offset:src_offset + n] = torch.from_numpy(data.copy()).cuda()  # (a copy: the input may be a read-only view)
    src = buf[src_offset:src_offset + n]
    nblocks = rcx.block_count(n, block)
    dst = torch.zeros(rcx.encode_bound(n, block, coder), dtype=torch.uint8, device="cuda")
    offs = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(src, block, dst, offs, coder=coder)
    ctx.sync_status()
    offsets = offs.cpu().numpy().astype(np.uint64)
    return dst[: int(offsets[-1])].cpu().numpy(), offsets, (dst, offs)


def gpu_decode(ctx, payload, offsets, n, block, dst_offset=0, comp_offset=0, coder=0):
    comp = torch.zeros(len(payload) + comp_offset + 16, dtype=torch.uint8, device="cuda")
    comp[comp_offset:comp_offset + len(payload)] = torch.from_numpy(np.ascontiguousarray(payload)).cuda()
    offs = torch.from_numpy(np.asarray(offsets).astype(np.int64)).cuda()
    out = torch.zeros(n + dst_offset + 16, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks_device(comp[comp_offset:], len(payload), offs, n, block, out[dst_offset:], coder=coder)
    st, bad = ctx.sync_status(raise_on_error=False)
    return out[dst_offset:dst_offset + n].cpu().numpy(), st, bad


def assert_same_blocks(payload, offsets, slots, sizes):
    assert np.array_equal(np.diff(offsets.astype(np.int64)), sizes.astype(np.int64)), "per-block sizes differ"
    for b in range(len(sizes)):
        got = payload[int(offsets[b]): int(offsets[b + 1])]
        assert np.array_equal(got, slots[b, : int(sizes[b])]), f"block {b} differs"


def test_golden_block_tables(ctx, golden):
    for t in golden["blocks"]:
        coder = 0 if t["coder"] == "adaptive" else 1
        data = workloads.by_name(t["workload"], t["n"], t["seed"])
        assert hashlib.sha256(data.tobytes()).hexdigest() == t["input_sha256"]
        payload, offsets, _ = gpu_encode(ctx, data, t["block"], coder=coder)
        assert [int(x) for x in np.diff(offsets.astype(np.int64))] == t["sizes"], (t["workload"], t["block"])
        fnv = ["%016x" % oracle_lib.fnv1a64(payload[int(offsets[b]): int(offsets[b + 1])]) for b in range(len(t["sizes"]))]
        assert fnv == t["fnv1a64"], (t["workload"], t["block"])
        assert int(offsets[-1]) == t["total"]
        back, st, _ = gpu_decode(ctx, payload, offsets, t["n"], t["block"], coder=coder)
        assert st == 0 and np.array_equal(back, data)


@pytest.mark.parametrize("block", [16, 48, 100, 1000, 4096, 16384, 65536, 65552, 262144, 1 << 20])
def test_ragged_sizes_match_oracle(ctx, oracle, block):
    rs = np.random.RandomState(block)
    for trial in range(3):
        nblocks = int(rs.randint(1, 200 if block <= 4096 else (70 if block <= 65552 else 5)))
        n = block * (nblocks - 1) + int(rs.randint(1, block + 1))  # last block ragged, may be 1 byte
        wl = ("uniform", "zipf", "runs", "canterbury")[trial % 4] if trial else "uniform"
        data = workloads.by_name(wl, n, 1000 + trial)
        slots, sizes = oracle.encode_blocks(data, block, threads=8)
        off = (0, 1, 5)[trial]  # the source need not be 16-byte aligned
        payload, offsets, _ = gpu_encode(ctx, data, block, src_offset=off)
        assert_same_blocks(payload, offsets, slots, sizes)
        back, st, _ = gpu_decode(ctx, payload, offsets, n, block, dst_offset=off, comp_offset=(0, 3, 2)[trial])
        assert st == 0 and np.array_equal(back, data)


def test_edge_inputs(ctx, oracle):
    files = workloads.canterbury_files()
    cases = [
        (np.full(300000, 255, np.uint8), 65536), (np.zeros(70000, np.uint8), 65536), (np.full(1 << 20, 65, np.uint8), 1 << 20),
        (np.arange(256, dtype=np.uint8).repeat(3), 768), (np.frombuffer(files["ptt5"], np.uint8), 65536),
        (np.frombuffer(files["kennedy.xls"], np.uint8), 1 << 20), (np.array([7], np.uint8), 65536),
        (np.frombuffer(files["alice29.txt"], np.uint8), 1 << 20),  # BASELINE config 1: alice29 as ONE block
    ]
    for data, block in cases:
        slots, sizes = oracle.encode_blocks(data, block, threads=8)
        payload, offsets, _ = gpu_encode(ctx, data, block)
        assert_same_blocks(payload, offsets, slots, sizes)
        back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block)
        assert st == 0 and np.array_equal(back, data)
    # alice29 single block: the README's pinned size (README.md:36, 0.573000 * 152089)
    data = np.frombuffer(files["alice29.txt"], np.uint8)
    payload, offsets, _ = gpu_encode(ctx, data, 1 << 20)
    assert len(payload) == 87147


def test_empty_input(ctx):
    from cpprcoder_amd import rcx
    src = torch.zeros(16, dtype=torch.uint8, device="cuda")[:0]
    dst = torch.zeros(64, dtype=torch.uint8, device="cuda")
    offs = torch.full((1,), 77, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(src, 65536, dst, offs)
    ctx.sync_status()
    assert int(offs[0]) == 0
    payload, offsets = ctx.encode_blocks(np.zeros(0, np.uint8), 65536)
    assert len(payload) == 0 and list(offsets) == [0]
    assert len(ctx.decode_blocks(payload, offsets, 65536)) == 0


def test_gpu_decodes_reference_streams_and_back(ctx, oracle):
    # streams produced by the CPU coder decode on the GPU, and GPU streams decode on the CPU coder
    data = workloads.canterbury_tiled(3_000_000)
    block = 65536
    slots, sizes = oracle.encode_blocks(data, block, threads=8)
    payload, offsets = oracle.compact(slots, sizes)
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block)
    assert st == 0 and np.array_equal(back, data)
    gp, go, _ = gpu_encode(ctx, data, block)
    gslots = np.zeros_like(slots)
    for b in range(len(sizes)):
        gslots[b, : int(sizes[b])] = gp[int(go[b]): int(go[b + 1])]
    cpu_back, ok = oracle.decode_blocks(gslots, sizes, block, len(data), threads=8)
    assert ok and np.array_equal(cpu_back, data)


def test_host_buffer_entry_points(ctx, oracle):
    data = workloads.zipf(1_000_003, 4)
    payload, offsets = ctx.encode_blocks(data, 65536)
    slots, sizes = oracle.encode_blocks(data, 65536, threads=8)
    assert_same_blocks(payload, offsets, slots, sizes)
    assert np.array_equal(ctx.decode_blocks(payload, offsets, 65536), data)


def test_errors_are_reported_not_fatal(ctx, oracle):
    from cpprcoder_amd import rcx
    data = workloads.uniform(65536 * 4 + 100, 3)
    payload, offsets, _ = gpu_encode(ctx, data, 65536)
    # header of block 2 disagrees with the layout -> corrupt, first bad block = 2
    bad = payload.copy()
    bad[int(offsets[2])] ^= 1
    _, st, blk = gpu_decode(ctx, bad, offsets, len(data), 65536)
    assert st == rcx.E_CORRUPT and blk == 2
    # truncated stream: the last block loses its tail
    cut = offsets.copy()
    cut[-1] -= 40
    _, st, blk = gpu_decode(ctx, payload[: int(cut[-1])], cut, len(data), 65536)
    assert st == rcx.E_CORRUPT and blk == 4
    # an offset table that points past the compressed buffer is not followed
    _, st, blk = gpu_decode(ctx, payload[:-100], offsets, len(data), 65536)
    assert st == rcx.E_CORRUPT and blk == 4
    _, st, blk = gpu_decode(ctx, payload[:-100], offsets, len(data), 65536, coder=0, comp_offset=3)
    assert st == rcx.E_CORRUPT and blk == 4
    # the context is still usable afterwards
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), 65536)
    assert st == 0 and np.array_equal(back, data)
    # destination too small for the compacted streams
    src = torch.from_numpy(data).cuda()
    dst = torch.zeros(100000, dtype=torch.uint8, device="cuda")
    offs = torch.zeros(6, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(src, 65536, dst, offs)
    st, _ = ctx.sync_status(raise_on_error=False)
    assert st == rcx.E_CAPACITY
    with pytest.raises(rcx.RcxError):
        ctx.encode_blocks_device(src, 8, dst, offs)  # block size out of range
    with pytest.raises(rcx.RcxError):
        ctx.encode_blocks_device(src, rcx.MAX_BLOCK + 16, dst, offs)  # a block the halving could reach (cpprcoder.h:1138)


def test_corrupt_payload_decodes_like_the_reference(ctx, oracle):
    """A damaged stream still decodes to SOME bytes in the reference (cpprcoder.h:900-917: no check
    beyond running out of input), including find()'s fall-through for a target at or past the
    total (cpprcoder.h:1220-1242).  The block entry points give the same bytes: the 4-lane kernel
    marks such blocks and the one-lane kernel decodes them again.  Every damaged stream is followed
    by random bytes so that neither decoder runs out of input."""
    block, nblocks = 4096, 96
    data = workloads.zipf(block * nblocks, 21)
    slots, sizes = oracle.encode_blocks(data, block, threads=8)
    rs = np.random.RandomState(5)
    pad = 3 * block
    big = np.zeros((nblocks, slots.shape[1] + pad), np.uint8)
    big_sizes = sizes.copy()
    hit = set()
    for b in range(nblocks):
        z = int(sizes[b])
        big[b, :z] = slots[b, :z]
        if b % 3:
            continue
        hit.add(b)
        big[b, z: z + pad] = rs.randint(0, 256, pad)
        big_sizes[b] = z + pad
        if b % 2 == 0:
            # the first renormalisation shifts in bytes 5..8: low >= 0xFFFFFF00 = 256 * (0xFFFFFF00 / 256),
            # the total's threshold -> the very first symbol is past the table
            big[b, 5:9] = 0xFF
        else:
            for _ in range(4):
                big[b, int(rs.randint(9, z))] ^= int(rs.randint(1, 256))
    want, ok = oracle.decode_blocks(big, big_sizes, block, len(data), threads=8)
    assert ok
    payload, offsets = oracle.compact(big, big_sizes)
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block)
    assert st == 0
    assert ctx.last_redo(nblocks) >= nblocks // 6  # at least the blocks damaged at their first symbol
    for b in range(nblocks):
        assert np.array_equal(back[b * block:(b + 1) * block], want[b * block:(b + 1) * block]), f"block {b}"
        if b not in hit:
            assert np.array_equal(back[b * block:(b + 1) * block], data[b * block:(b + 1) * block])
    assert not np.array_equal(want, data)


def test_carry_through_long_runs_of_ff(ctx, oracle):
    """A carry through a held run of 0xFF bytes (cpprcoder.h:767-800).  Runs shorter than what the five-wave
    encoder keeps in LDS are resolved there; longer ones make it hand the block to the one-lane kernel.  Either
    way the stream is the reference's, and it decodes."""
    import carry_runs
    block = 4096
    runs = [0, 3, 20, 30, 40, 70, 300, 0, 0, 45, 0, 1000]
    parts = [carry_runs.carry_run_block(block, r, 100 + i) if r else workloads.uniform(block, 100 + i) for i, r in enumerate(runs)]
    data = np.concatenate(parts)
    payload, offsets, _ = gpu_encode(ctx, data, block)
    redone = ctx.last_redo(len(runs))
    slots, sizes = oracle.encode_blocks(data, block, threads=4)
    assert_same_blocks(payload, offsets, slots, sizes)
    assert redone >= sum(r > 64 for r in runs), redone   # those cannot have been resolved in the 32-byte margin
    assert redone <= sum(r > 24 for r in runs), redone   # and short runs must not take the slow path
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block)
    assert st == 0 and np.array_equal(back, data)
    assert ctx.last_redo(len(runs)) == 0                  # a valid stream never takes the decoder's slow path
    # ordinary data: nothing is handed over
    plain = workloads.zipf(64 * 65536, 7)
    p2, o2, _ = gpu_encode(ctx, plain, 65536)
    assert ctx.last_redo(64) == 0
    assert np.array_equal(gpu_decode(ctx, p2, o2, len(plain), 65536)[0], plain)


def test_carries_at_many_output_positions(ctx, oracle):
    """The multi-wave encoder's writer mirrors a window of the newest eight bytes into a 256-byte ring and lets a carry that
    runs through more than four bytes go on in the ring (csrc/rcx_oct.hpp StagedWriter): runs of 4 .. 40 held 0xFF bytes
    that begin at many places of a block -- whatever their position relative to the ring's wrap, a word boundary, a chunk
    of 16 symbols -- must give the reference's stream."""
    import carry_runs
    block = 4096
    parts, runs = [], []
    for i, run in enumerate([4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 24, 33, 40]):
        for k, after in enumerate([0, 150, 260, 515, 770, 1290, 2000, 3000]):
            try:
                parts.append(carry_runs.carry_run_block(block, run, 9000 + 100 * i + k, after))
                runs.append(run)
            except AssertionError:  # (no straddle behind `after` with this seed: the next one)
                continue
    assert len(parts) >= 80
    data = np.concatenate(parts)
    payload, offsets, _ = gpu_encode(ctx, data, block)
    slots, sizes = oracle.encode_blocks(data, block, threads=8)
    assert_same_blocks(payload, offsets, slots, sizes)
    assert ctx.last_redo(len(runs)) <= sum(r > 24 for r in runs)
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block)
    assert st == 0 and np.array_equal(back, data)


def test_static_carry_through_long_runs_of_ff(ctx, oracle):
    """The same for the static coder's three-wave encoder (rcx_enc_static3_k) and its second pass."""
    import carry_runs
    block = 8192

    def make(run, i):
        for seed in range(50):
            try:
                return carry_runs.carry_run_block_static(block, run, 1000 * i + seed)
            except AssertionError:
                pass
        raise RuntimeError("no carry run found")

    runs = [0, 5, 20, 60, 100, 0, 45]
    parts = [make(r, i) if r else workloads.uniform(block, 300 + i) for i, r in enumerate(runs)]
    data = np.concatenate(parts)
    payload, offsets, _ = gpu_encode(ctx, data, block, coder=1)
    redone = ctx.last_redo(len(runs))
    slots, sizes = oracle.encode_blocks(data, block, coder=1, threads=4)
    assert_same_blocks(payload, offsets, slots, sizes)
    assert sum(r > 64 for r in runs) <= redone <= sum(r > 24 for r in runs), redone
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block, coder=1)
    assert st == 0 and np.array_equal(back, data) and ctx.last_redo(len(runs)) == 0


def test_bursts_of_very_improbable_symbols(ctx, oracle):
    """After ~2^20 equal bytes every other byte value costs 20 bits: 2.5 output bytes per symbol for a while.
    That is more than the decoder's fast input top-up supplies (8 dwords per 16 symbols: its synchronous refill
    runs) and more than one 16-byte piece per chunk for the encoder's drain."""
    block = 1 << 20
    def one(fill, seed):
        rs = np.random.RandomState(seed)
        d = np.full(block, fill, np.uint8)
        others = np.array([v for v in range(256) if v != fill], np.uint8)
        tail = np.concatenate([rs.permutation(others) for _ in range(24)])
        d[block - len(tail):] = tail
        d[5000:5000 + 255] = rs.permutation(others)  # and once early, at 13 bits per symbol
        return d
    data = np.concatenate([one(65, 1), one(0, 2), one(255, 3), workloads.zipf(block, 4)])
    payload, offsets, _ = gpu_encode(ctx, data, block)
    slots, sizes = oracle.encode_blocks(data, block, threads=4)
    assert_same_blocks(payload, offsets, slots, sizes)
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block)
    assert st == 0 and np.array_equal(back, data)
    assert ctx.last_redo(4) == 0


def test_single_stream_semantics(ctx, oracle, golden):
    """rcx_stream_encode / rcx_stream_decode == the reference's initialize+encode / initialize+decode
    on a MemoryStream of the given capacity (cpprcoder.h:678-720, 859-924, 1047-1054)."""
    for k in golden["kat"]["kat"]:
        v = bytes.fromhex(k["input_hex"])
        st, rq, out = ctx.stream_encode(v)
        assert (st, rq, out.hex()) == (0, 0, k["adaptive_hex"])
        st, rq, back = ctx.stream_decode(out, max(len(v), 16))
        assert st == 0 and back == (v if v else b"\x00")  # cpprcoder.h:912 quirk for the empty stream
    def r16(cap):  # what MemoryStream(cap).capacity() is (cpprcoder.h:975)
        return 16 if cap <= 0 else (cap + 15) & ~15

    u64k = workloads.uniform(65536, 12345)
    st, rq, out = ctx.stream_encode(u64k, sink_capacity=65536)  # {Pending, 112}
    pin = golden["kat"]["pins"]["overflow_uniform64k_into_65536"]
    assert [st, rq] == pin["status"] == [1, 112] and len(out) == pin["size"]
    assert hashlib.sha256(out).hexdigest() == pin["prefix_sha256"]
    rs = np.random.RandomState(77)
    for _ in range(25):
        n = int(rs.randint(1, 4000))
        data = workloads.by_name(("uniform", "zipf", "runs")[int(rs.randint(3))], n, int(rs.randint(1 << 30)))
        full = oracle.adaptive_encode(data)[2]
        for cap in (int(rs.randint(16, full + 32)), max(16, full - int(rs.randint(0, 12)))):
            (rst, rrq), rout, rsize = oracle.adaptive_encode(data, sink_capacity=cap)
            st, rq, out = ctx.stream_encode(data, sink_capacity=r16(cap))
            assert (st, rq, len(out)) == (rst, rrq, rsize) and out[: len(rout)] == rout
    pins = golden["kat"]["pins"]
    st, rq, out = ctx.stream_decode(b"\x01\x00\x00", 16)
    assert [st, rq] == pins["decode_short_input"]["status"] and out == b""
    for key in ("decode_junk_64", "decode_allff_64"):
        st, rq, out = ctx.stream_decode(bytes.fromhex(pins[key]["input_hex"]), 64)
        assert [st, rq] == pins[key]["status"] and out.hex() == pins[key]["out_hex"]
    comp = oracle.adaptive_encode(u64k[:3000])[1]
    st, rq, out = ctx.stream_decode(comp[:1500], 3000)
    assert [st, rq] == pins["decode_truncated"]["status"] and len(out) == pins["decode_truncated"]["size"]
    assert hashlib.sha256(out).hexdigest() == pins["decode_truncated"]["out_sha256"]
    st, rq, out = ctx.stream_decode(comp, r16(1000))
    assert [st, rq] == pins["decode_sink_full"]["status"] and len(out) == pins["decode_sink_full"]["size"]
    assert hashlib.sha256(out[:1000]).hexdigest() == pins["decode_sink_full"]["out_sha256"]  # the fixture hashed the first 1000 bytes


FULL_SIZE_RATIOS = {  # SURVEY.md section 8(d) / the reference build on these very bytes (profiles/r02_coder_soak.txt)
    ("uniform", 0): 1.001814, ("canterbury", 0): 0.456671, ("zipf", 0): 0.779815,
}


@pytest.mark.parametrize("workload", ["uniform", "canterbury", "zipf"])
def test_full_size_every_block(ctx, oracle, workload):
    """BASELINE.json configs[1] (the std::mt19937(12345) GiB), configs[2] (the Canterbury GiB) and one shard of
    configs[3] (the Zipf GiB of rank 0), 64 KiB blocks, at full size: EVERY one of the 16384 block streams byte for byte
    against the oracle (all host threads), for the adaptive coder and once each for the static and the two rANS coders, plus
    the round trip -- the reference harness's own every-byte compare (test/main.cpp:357-361) on BASELINE's own buffers."""
    import os
    from cpprcoder_amd import rcx
    n, block = 1 << 30, 65536
    nblocks = n // block
    data = workloads.by_name(workload, n, 12345)
    src = torch.from_numpy(data).cuda()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    offs = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
    threads = min(os.cpu_count() or 16, 128)
    for coder in (rcx.CODER_ADAPTIVE, rcx.CODER_STATIC, rcx.CODER_RANS, rcx.CODER_RANS8):
        dst = torch.empty(rcx.encode_bound(n, block, coder), dtype=torch.uint8, device="cuda")
        ctx.encode_blocks_device(src, block, dst, offs, coder=coder)
        ctx.sync_status()
        offsets = offs.cpu().numpy()
        total = int(offsets[-1])
        payload = dst[:total].cpu().numpy()
        slots, sizes = oracle.encode_blocks(data, block, coder=coder, threads=threads)
        assert np.array_equal(np.diff(offsets), sizes.astype(np.int64)), f"coder {coder}: per-block sizes differ from the oracle"
        want, _ = oracle.compact(slots, sizes)
        del slots
        assert np.array_equal(payload, want), f"coder {coder}: block {int(np.searchsorted(offsets, int(np.argmax(payload != want)), side='right')) - 1} differs"
        del want, payload
        if (workload, coder) in FULL_SIZE_RATIOS:
            assert abs(total / n - FULL_SIZE_RATIOS[(workload, coder)]) < 1e-6
        out.zero_()
        ctx.decode_blocks_device(dst, total, offs, n, block, out, coder=coder)
        ctx.sync_status()
        assert torch.equal(out, src), f"coder {coder}: round trip"
        if coder in (rcx.CODER_ADAPTIVE, rcx.CODER_STATIC):
            assert ctx.last_redo(nblocks) == 0  # (no block of a valid stream is left to the one-lane kernels)
        if coder == rcx.CODER_ADAPTIVE:  # every header says 65536 and every stream ends where the next begins
            heads = dst[offs[:-1].unsqueeze(1) + torch.arange(4, device="cuda").unsqueeze(0)].cpu().numpy().astype(np.uint32)
            assert np.all(heads[:, 0] + (heads[:, 1] << 8) + (heads[:, 2] << 16) + (heads[:, 3] << 24) == block)
        del dst


# ---------------------------------------------------------------------------
# Static (two-pass) coder, RangeEncoder<T> (cpprcoder.h:321-619)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("block", [16, 100, 4096, 65536, 65552, 262144])
def test_static_ragged_sizes_match_oracle(ctx, oracle, block):
    rs = np.random.RandomState(block + 1)
    for trial in range(3):
        nblocks = int(rs.randint(1, 150 if block <= 4096 else (70 if block <= 65552 else 5)))
        n = block * (nblocks - 1) + int(rs.randint(1, block + 1))
        wl = ("uniform", "zipf", "runs", "canterbury")[trial % 4] if trial else "canterbury"
        data = workloads.by_name(wl, n, 2000 + trial)
        slots, sizes = oracle.encode_blocks(data, block, coder=1, threads=8)
        off = (0, 1, 5)[trial]
        payload, offsets, _ = gpu_encode(ctx, data, block, src_offset=off, coder=1)
        assert_same_blocks(payload, offsets, slots, sizes)
        back, st, _ = gpu_decode(ctx, payload, offsets, n, block, dst_offset=off, comp_offset=(0, 3, 2)[trial], coder=1)
        assert st == 0 and np.array_equal(back, data)


def test_static_edge_inputs(ctx, oracle, golden):
    files = workloads.canterbury_files()
    cases = [
        (np.full(65535, 65, np.uint8), 65536), (np.full(65536, 65, np.uint8), 65536),   # the 16-bit squeeze (cpprcoder.h:549-555)
        (np.full(65537 + 65536, 65, np.uint8), 1 << 20), (np.full(1 << 20, 200, np.uint8), 1 << 20),
        (np.concatenate([np.full(70000, 1, np.uint8), workloads.zipf(200000, 3), np.full(140000, 1, np.uint8)]), 1 << 20),
        (np.arange(256, dtype=np.uint8).repeat(3), 768), (np.array([7], np.uint8), 65536), (np.array([0, 255], np.uint8), 16),
        (np.frombuffer(files["ptt5"], np.uint8), 1 << 20), (np.frombuffer(files["kennedy.xls"], np.uint8), 1 << 20),
        (np.frombuffer(files["alice29.txt"], np.uint8), 1 << 20),
    ]
    for data, block in cases:
        slots, sizes = oracle.encode_blocks(data, block, coder=1, threads=8)
        payload, offsets, _ = gpu_encode(ctx, data, block, coder=1)
        assert_same_blocks(payload, offsets, slots, sizes)
        back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block, coder=1)
        assert st == 0 and np.array_equal(back, data)
    # README.md:20-30: the published static sizes, each file as one block
    for name in workloads.CANTERBURY_ORDER:
        data = np.frombuffer(files[name], np.uint8)
        payload, offsets, _ = gpu_encode(ctx, data, 1 << 20, coder=1)
        assert len(payload) == golden["kat"]["canterbury"][name]["static_size"]
        assert hashlib.sha256(payload.tobytes()).hexdigest() == golden["kat"]["canterbury"][name]["static_sha256"]


def test_static_corrupt_streams(ctx, oracle):
    from cpprcoder_amd import rcx
    data = workloads.zipf(65536 * 3 + 500, 8)
    payload, offsets, _ = gpu_encode(ctx, data, 65536, coder=1)
    cut = offsets.copy()
    cut[-1] -= 30  # truncated last block
    _, st, blk = gpu_decode(ctx, payload[: int(cut[-1])], cut, len(data), 65536, coder=1)
    assert st == rcx.E_CORRUPT and blk == 3
    bad = payload.copy()
    bad[int(offsets[1]) + 4: int(offsets[1]) + 516] = 0  # block 1: all counts zero -> total 0
    _, st, blk = gpu_decode(ctx, bad, offsets, len(data), 65536, coder=1)
    assert st == rcx.E_CORRUPT and blk == 1
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), 65536, coder=1)
    assert st == 0 and np.array_equal(back, data)
    # streams made by the CPU coder decode on the GPU
    slots, sizes = oracle.encode_blocks(data, 65536, coder=1, threads=8)
    p2, o2 = oracle.compact(slots, sizes)
    back, st, _ = gpu_decode(ctx, p2, o2, len(data), 65536, coder=1)
    assert st == 0 and np.array_equal(back, data)


def test_superseded_kernels_in_the_diagnostic_build():
    """csrc/variants/ (8 lanes per block, the four-wave encoder) is not in librcx.so any more; the diagnostic library that
    carries it (build.py variants=True) must still produce the oracle's bytes -- checked in a process of its own, because a
    process binds one library."""
    import os
    import subprocess
    import sys
    from cpprcoder_amd import build
    lib = build.build(variants=True)
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, RCX_LIBRARY=lib)
    for k in ("RCX_ENC_VARIANT", "RCX_LANES_PER_BLOCK", "RCX_ENC_LANES", "RCX_DEC_QUADS"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(here, "variants_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "variants ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_every_kernel_variant_is_bit_identical(oracle):
    """The selectable encode kernels of the product (one wave per 64 blocks / the five-wave split) and decode kernels
    (1 or 4 lanes per block) and their launch shapes must all produce the oracle's bytes."""
    import os
    from cpprcoder_amd import rcx
    data = workloads.canterbury_tiled(65536 * 37 + 4321)
    saved = {k: os.environ.get(k) for k in ("RCX_ENC_VARIANT", "RCX_LANES_PER_BLOCK", "RCX_ENC_LANES", "RCX_DEC_QUADS")}
    try:
        for block in (4096, 65536):
            slots, sizes = oracle.encode_blocks(data, block, threads=8)
            ref_payload, ref_offsets = oracle.compact(slots, sizes)
            for enc in ("0", "3"):
                for dec in ("1", "4"):
                    os.environ["RCX_ENC_VARIANT"], os.environ["RCX_LANES_PER_BLOCK"] = enc, dec
                    c = rcx.Context(0)
                    payload, offsets, _ = gpu_encode(c, data, block)
                    assert np.array_equal(payload, ref_payload) and np.array_equal(offsets, ref_offsets), (block, enc)
                    back, st, _ = gpu_decode(c, payload, offsets, len(data), block)
                    assert st == 0 and np.array_equal(back, data), (block, dec)
                    c.close()
            # launch shapes (rcx_api.hip: encode_lanes / decode_quads): fewer blocks per workgroup / per wave
            for lanes, quads in (("1", "1"), ("8", "4"), ("32", "8"), ("5", "2")):
                os.environ["RCX_ENC_VARIANT"], os.environ["RCX_LANES_PER_BLOCK"] = "3", "4"
                os.environ["RCX_ENC_LANES"], os.environ["RCX_DEC_QUADS"] = lanes, quads
                c = rcx.Context(0)
                for coder, want_p, want_o in ((0, ref_payload, ref_offsets), (1, None, None)):
                    if coder == 1:
                        s_slots, s_sizes = oracle.encode_blocks(data, block, coder=1, threads=8)
                        want_p, want_o = oracle.compact(s_slots, s_sizes)
                    payload, offsets, _ = gpu_encode(c, data, block, coder=coder)
                    assert np.array_equal(payload, want_p) and np.array_equal(offsets, want_o), (block, lanes, coder)
                    back, st, _ = gpu_decode(c, payload, offsets, len(data), block, coder=coder)
                    assert st == 0 and np.array_equal(back, data), (block, quads, coder)
                c.close()
                os.environ.pop("RCX_ENC_LANES"), os.environ.pop("RCX_DEC_QUADS")
            # static coder: one-wave / three-wave encoder x one-lane / four-lane decoder
            s_slots, s_sizes = oracle.encode_blocks(data, block, coder=1, threads=8)
            s_payload, s_offsets = oracle.compact(s_slots, s_sizes)
            for enc in ("0", "3"):
                for dec in ("1", "4"):
                    os.environ["RCX_ENC_VARIANT"], os.environ["RCX_LANES_PER_BLOCK"] = enc, dec
                    c = rcx.Context(0)
                    payload, offsets, _ = gpu_encode(c, data, block, coder=1)
                    assert np.array_equal(payload, s_payload) and np.array_equal(offsets, s_offsets), ("static", block, enc)
                    back, st, _ = gpu_decode(c, payload, offsets, len(data), block, coder=1)
                    assert st == 0 and np.array_equal(back, data), ("static", block, dec)
                    c.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_streams_and_graph_replay(ctx, oracle):
    """The device entry points only enqueue work on the caller's stream (no allocation after
    rcx_ctx_reserve, no host sync), so they run on a side stream and can be captured into a HIP graph."""
    from cpprcoder_amd import rcx
    n, block = 65536 * 64 + 100, 65536
    data = workloads.zipf(n, 77)
    slots, sizes = oracle.encode_blocks(data, block, threads=8)
    ref_payload, ref_offsets = oracle.compact(slots, sizes)
    src = torch.from_numpy(data).cuda()
    nblocks = rcx.block_count(n, block)
    dst = torch.zeros(rcx.encode_bound(n, block), dtype=torch.uint8, device="cuda")
    offs = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ctx.reserve(n, block)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ctx.encode_blocks_device(src, block, dst, offs)
        ctx.decode_blocks_device(dst, dst.numel(), offs, n, block, out)
        ctx.sync_status()
    side.synchronize()
    assert np.array_equal(offs.cpu().numpy().astype(np.uint64), ref_offsets)
    assert np.array_equal(dst[: int(ref_offsets[-1])].cpu().numpy(), ref_payload) and torch.equal(out, src)
    # capture once, replay on new input
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ctx.encode_blocks_device(src, block, dst, offs)
        ctx.decode_blocks_device(dst, dst.numel(), offs, n, block, out)
    data2 = workloads.canterbury_tiled(n)
    src.copy_(torch.from_numpy(data2).cuda())
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    ctx.sync_status()
    s2, z2 = oracle.encode_blocks(data2, block, threads=8)
    p2, o2 = oracle.compact(s2, z2)
    assert np.array_equal(offs.cpu().numpy().astype(np.uint64), o2)
    assert np.array_equal(dst[: int(o2[-1])].cpu().numpy(), p2) and torch.equal(out, src)


def test_many_small_random_buffers(ctx, oracle):
    rs = np.random.RandomState(4242)
    for _ in range(120):
        block = int(rs.choice([16, 17 * 16, 256, 1000, 4096]))
        n = int(rs.randint(1, 40000))
        alpha = int(rs.choice([1, 2, 3, 16, 200, 256]))
        data = rs.randint(0, alpha, size=n).astype(np.uint8)
        coder = int(rs.randint(4))  # adaptive, static, rANS one state, rANS eight states
        slots, sizes = oracle.encode_blocks(data, block, coder=coder, threads=4)
        payload, offsets, _ = gpu_encode(ctx, data, block, coder=coder, src_offset=int(rs.randint(4)))
        assert_same_blocks(payload, offsets, slots, sizes)
        back, st, _ = gpu_decode(ctx, payload, offsets, n, block, coder=coder, comp_offset=int(rs.randint(16)))
        assert st == 0 and np.array_equal(back, data), (block, n, alpha, coder)


# ---------------------------------------------------------------------------
# Past 1 MiB: big blocks (no halving up to RCX_MAX_BLOCK = 2^24 - 256) and long single streams through the
# halving at total = 2^24 (cpprcoder.h:1138-1176); fixtures from the reference (tests/golden/make_golden_long.py)
# ---------------------------------------------------------------------------
def test_blocks_past_one_mebibyte(ctx, golden):
    for t in golden["long"]["blocks"]:
        coder = 0 if t["coder"] == "adaptive" else 1
        data = workloads.by_name(t["workload"], t["n"], t["seed"])
        assert hashlib.sha256(data.tobytes()).hexdigest() == t["input_sha256"]
        payload, offsets, _ = gpu_encode(ctx, data, t["block"], coder=coder)
        assert [int(x) for x in np.diff(offsets.astype(np.int64))] == t["sizes"], (t["workload"], t["block"], t["coder"])
        fnv = ["%016x" % oracle_lib.fnv1a64(payload[int(offsets[b]): int(offsets[b + 1])]) for b in range(len(t["sizes"]))]
        assert fnv == t["fnv1a64"], (t["workload"], t["block"], t["coder"])
        back, st, _ = gpu_decode(ctx, payload, offsets, t["n"], t["block"], coder=coder)
        assert st == 0 and np.array_equal(back, data)


def test_block_of_the_largest_size(ctx, golden):
    """One block of RCX_MAX_BLOCK = 2^24 - 256 symbols: the last size whose table is never halved, through the
    many-lane kernels (divisor table built on the device up to total 2^24)."""
    from cpprcoder_amd import rcx
    g = golden["long"]["adaptive"]["zipf(NO_HALVING,4)"]
    data = workloads.zipf(rcx.MAX_BLOCK, 4)
    assert hashlib.sha256(data.tobytes()).hexdigest() == g["input_sha256"]
    payload, offsets, _ = gpu_encode(ctx, data, rcx.MAX_BLOCK)
    assert len(payload) == g["size"] and hashlib.sha256(payload.tobytes()).hexdigest() == g["sha256"]
    back, st, _ = gpu_decode(ctx, payload, offsets, len(data), rcx.MAX_BLOCK)
    assert st == 0 and np.array_equal(back, data)


def test_long_single_streams(ctx, golden):
    """rcx_stream_encode / rcx_stream_decode on streams of 2 MiB ... 2^24 + 70000 symbols: the reference's bytes
    (cpprcoder.h:697-720 has no size cap), including the halve-and-resum step (cpprcoder.h:1138-1176) and, for the
    static coder, count()'s second rescale (cpprcoder.h:561-570)."""
    import test_oracle_golden
    adaptive, static = test_oracle_golden.long_inputs()
    for label, g in golden["long"]["adaptive"].items():
        v = adaptive[label]()
        st, rq, comp = ctx.stream_encode(v)
        assert (st, rq, len(comp)) == (0, 0, g["size"]), label
        assert hashlib.sha256(comp).hexdigest() == g["sha256"], label
        st, rq, back = ctx.stream_decode(comp, len(v))
        assert (st, rq) == (0, 0) and back == v.tobytes(), label
    for label, g in golden["long"]["static"].items():
        v = static[label]()
        st, rq, comp = ctx.stream_encode(v, coder=1)
        assert st == 0 and len(comp) == g["size"] and hashlib.sha256(comp).hexdigest() == g["sha256"], label
        st, rq, back = ctx.stream_decode(comp, len(v), coder=1)
        assert st == 0 and back == v.tobytes(), label
    # a long stream into a sink that fills: {Pending, remaining} as cpprcoder.h:708-711
    v = adaptive["uniform(2MiB+77,3)"]()
    st, rq, out = ctx.stream_encode(v, sink_capacity=1 << 20)
    assert st == 1 and 0 < rq < len(v) and len(out) == 1 << 20


def test_stream_encode_respects_dst_cap(ctx, oracle):
    """include/rcx.h: dst holds dst_cap bytes; a result that does not fit is reported, never written.  (Context.stream_encode
    hands over exactly the documented size and checks canary bytes behind it on every call of this suite.)"""
    from cpprcoder_amd import rcx
    data = workloads.zipf(5000, 3)
    (st0, _), ref, size = oracle.adaptive_encode(data)
    for cap in (size - 20, size - 4, size, size + 100):  # sinks around "everything fits through writeByte"
        cap16 = (cap + 15) & ~15  # what MemoryStream(cap).capacity() is (cpprcoder.h:975)
        st, rq, out = ctx.stream_encode(data, sink_capacity=cap16)
        (rst, rrq), rout, rsize = oracle.adaptive_encode(data, sink_capacity=cap)
        assert (st, rq, len(out)) == (rst, rrq, rsize) and out[: len(rout)] == rout
    st, rq, out = ctx.stream_encode(data, sink_capacity=size, dst_cap=size - 1)
    assert st == rcx.E_CAPACITY and out == b""
    st, rq, out = ctx.stream_encode(data, sink_capacity=64, dst_cap=63)  # the pending path copies 64 bytes
    assert st == rcx.E_CAPACITY
    st, rq, out = ctx.stream_encode(data, sink_capacity=64, dst_cap=64)
    assert st == rcx.PENDING and len(out) == 64 and out == ref[:64]
    st, rq, out = ctx.stream_encode(data, coder=1, dst_cap=600)
    assert st == rcx.E_CAPACITY


def test_static_histogram_past_65535_symbols(ctx, oracle):
    """count() (cpprcoder.h:543-571) in blocks longer than 65535 symbols: the calm case (no count can reach 0xFFFF:
    all three waves count with atomics) and the order-dependent 16-bit squeeze (:549-555) in the middle of a block."""
    block = 262144
    calm = workloads.uniform(block * 6, 31)
    text = workloads.canterbury_tiled(block * 5 + 1000)
    squeeze = np.concatenate([workloads.runs(block * 3, 5), np.full(block, 7, np.uint8),
                              np.concatenate([np.full(70000, 1, np.uint8), workloads.zipf(block - 70000 - 65000, 3), np.full(65000, 1, np.uint8)])])
    for data in (calm, text, squeeze):
        slots, sizes = oracle.encode_blocks(data, block, coder=1, threads=8)
        payload, offsets, _ = gpu_encode(ctx, data, block, coder=1)
        assert_same_blocks(payload, offsets, slots, sizes)
        back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block, coder=1)
        assert st == 0 and np.array_equal(back, data)


def test_resumable_decoder_fed_in_pieces(ctx, oracle, golden):
    """rcx_dstream_*: AdaptiveRangeDecoder<T>::decode called piece by piece (cpprcoder.h:872-924).  The coder state stays
    on the GPU between calls; whatever the split, the symbols are the reference's, {Pending, remaining} is reported
    while the input is short (cpprcoder.h:901-903), and a stream fed in many pieces costs what its pieces cost."""
    import time
    from cpprcoder_amd import rcx
    files = workloads.canterbury_files()
    cases = [b"", b"hello world", workloads.zipf(20000, 5).tobytes(), workloads.uniform(70000, 3).tobytes(), files["fields.c"]]
    for v in cases:
        comp = oracle.adaptive_encode(v)[1]
        want = v if v else b"\x00"  # cpprcoder.h:912: a stream declaring 0 bytes yields one
        for piece in (8, 9, 100, 4096, len(comp)):
            d = ctx.dstream()
            out, at, st, rq = b"", 0, rcx.PENDING, 0
            while at < len(comp) and st == rcx.PENDING:
                st, rq, got = d.decode(comp[at: at + piece], 1 << 20)
                at += piece
                out += got
                assert want.startswith(out)
                if st == rcx.PENDING:
                    assert rq == len(v) - len(out)
            assert st == rcx.OK and out == want, (len(v), piece)
            d.close()
    # fewer than 8 bytes first: nothing is kept, 8 are asked for (cpprcoder.h:877-880)
    d = ctx.dstream()
    comp = oracle.adaptive_encode(cases[2])[1]
    assert d.decode(comp[:5], 100)[:2] == (rcx.PENDING, 8)
    st, rq, got = d.decode(comp, 100)  # a sink with room for 100: the caller comes back for more with no new bytes
    assert (st, rq, got) == (rcx.PENDING, len(cases[2]) - 100, cases[2][:100])
    rest = b""
    while st == rcx.PENDING:
        st, rq, got = d.decode(b"", 3000)
        rest += got
    assert st == rcx.OK and cases[2][:100] + rest == cases[2]
    d.close()
    # damaged input decodes like the reference (find()'s fall-through included)
    for key in ("decode_junk_64", "decode_allff_64"):
        pin = golden["kat"]["pins"][key]
        d = ctx.dstream()
        st, rq, got = d.decode(bytes.fromhex(pin["input_hex"]), 64)
        assert [st, rq] == pin["status"] and got.hex() == pin["out_hex"]
        d.close()
    # 2 MiB in 4 KiB pieces: 500+ calls, each decoding only its own symbols
    big = workloads.zipf(2 << 20, 77).tobytes()
    comp = oracle.adaptive_encode(big)[1]
    d = ctx.dstream()
    t0, out = time.time(), []
    for at in range(0, len(comp), 4096):
        st, rq, got = d.decode(comp[at: at + 4096], 1 << 20)
        out.append(got)
    assert st == rcx.OK and b"".join(out) == big
    assert time.time() - t0 < 60
    d.close()
