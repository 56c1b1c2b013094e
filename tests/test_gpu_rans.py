"""GPU parity of the rANS coders (RCX_CODER_RANS / RCX_CODER_RANS8; cppans.h:497-649) through the C ABI: every block's
stream must equal what the reference's rANS::encode / encode_simd return for that block (fixtures made from the
reference build: tests/golden/rans.json; the oracle restates cppans.h and is itself pinned to that build), and
decode(encode(x)) == x (test/main.cpp:387-393)."""
import hashlib

import numpy as np
import pytest

import oracle_lib
from cpprcoder_amd import workloads
from test_gpu_parity import assert_same_blocks, gpu_decode, gpu_encode

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
CODERS = ((2, "rans"), (3, "rans8"))


@pytest.fixture(scope="module")
def ctx():
    from cpprcoder_amd import rcx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = rcx.Context(0)
    yield c
    c.close()


def test_rans_kats_through_the_stream_calls(ctx, golden):
    for k in golden["rans"]["kat"]:
        v = bytes.fromhex(k["input_hex"])
        for coder, key in CODERS:
            st, rq, comp = ctx.stream_encode(v, coder=coder)
            assert st == 0 and comp.hex() == k[key + "_hex"], (v, key)
            st, rq, back = ctx.stream_decode(comp, max(len(v), 16), coder=coder)
            assert st == 0 and back == v
    from cpprcoder_amd import rcx
    for coder, _ in CODERS:
        assert ctx.stream_encode(b"", coder=coder)[0] == rcx.ERROR            # cppans.h:501: 0 < src_size
        st, _, comp = ctx.stream_encode(b"abc" * 100, coder=coder)
        assert ctx.stream_decode(comp, 299, coder=coder)[0] == rcx.ERROR      # cppans.h:541 / :618: destination too small
        assert ctx.stream_decode(comp[:500], 300, coder=coder)[0] == rcx.ERROR


def test_rans_generated_and_canterbury_files(ctx, golden):
    import test_oracle_rans
    gens = test_oracle_rans.generated()
    for name, g in golden["rans"]["generated"].items():
        v = gens[name]()
        for coder, key in CODERS:
            st, _, comp = ctx.stream_encode(v, coder=coder)
            assert st == 0 and (len(comp), hashlib.sha256(comp).hexdigest()) == (g[key + "_size"], g[key + "_sha256"]), (name, key)
            st, _, back = ctx.stream_decode(comp, len(v), coder=coder)
            assert st == 0 and back == v.tobytes()
    files = workloads.canterbury_files()
    for name, g in golden["rans"]["canterbury"].items():
        v = np.frombuffer(files[name], np.uint8)
        for coder, key in CODERS:
            st, _, comp = ctx.stream_encode(v, coder=coder)
            assert st == 0 and (len(comp), hashlib.sha256(comp).hexdigest()) == (g[key + "_size"], g[key + "_sha256"]), (name, key)


def test_rans_golden_block_tables(ctx, golden):
    for t in golden["rans"]["blocks"]:
        coder = 2 if t["coder"] == "rans" else 3
        data = workloads.by_name(t["workload"], t["n"], t["seed"])
        assert hashlib.sha256(data.tobytes()).hexdigest() == t["input_sha256"]
        payload, offsets, _ = gpu_encode(ctx, data, t["block"], coder=coder)
        assert [int(x) for x in np.diff(offsets.astype(np.int64))] == t["sizes"], (t["workload"], t["block"], t["coder"])
        fnv = ["%016x" % oracle_lib.fnv1a64(payload[int(offsets[b]): int(offsets[b + 1])]) for b in range(len(t["sizes"]))]
        assert fnv == t["fnv1a64"], (t["workload"], t["block"], t["coder"])
        back, st, _ = gpu_decode(ctx, payload, offsets, t["n"], t["block"], coder=coder)
        assert st == 0 and np.array_equal(back, data)


@pytest.mark.parametrize("block", [16, 24, 100, 1000, 4096, 65536, 65552, 262144])
def test_rans_ragged_sizes_match_oracle(ctx, oracle, block):
    rs = np.random.RandomState(block + 7)
    for trial in range(3):
        nblocks = int(rs.randint(1, 150 if block <= 4096 else (40 if block <= 65552 else 5)))
        n = block * (nblocks - 1) + int(rs.randint(1, block + 1))  # last block ragged, may be 1 byte (fewer symbols than states)
        wl = ("uniform", "zipf", "runs", "canterbury")[trial % 4] if trial else "canterbury"
        data = workloads.by_name(wl, n, 3000 + trial)
        off = (0, 1, 5)[trial]  # the source need not be aligned
        for coder, _ in CODERS:
            slots, sizes = oracle.encode_blocks(data, block, coder=coder, threads=8)
            payload, offsets, _ = gpu_encode(ctx, data, block, src_offset=off, coder=coder)
            assert_same_blocks(payload, offsets, slots, sizes)
            back, st, _ = gpu_decode(ctx, payload, offsets, n, block, dst_offset=off, comp_offset=(0, 3, 2)[trial], coder=coder)
            assert st == 0 and np.array_equal(back, data)


def test_rans_edge_inputs(ctx, oracle):
    rs = np.random.RandomState(9)
    cases = [
        (np.full(300000, 255, np.uint8), 65536),   # one symbol: frequency 4096 -- encode_simd's word per symbol (cppans.h:357)
        (np.zeros(70000, np.uint8), 65536), (np.array([7], np.uint8), 65536), (np.arange(256, dtype=np.uint8).repeat(3), 768),
        (np.concatenate([np.zeros(70000, np.uint8), np.ones(1, np.uint8)]), 1 << 20),           # normalize()'s steal loop
        (np.concatenate([workloads.zipf(300000, 3), np.arange(256, dtype=np.uint8)]), 1 << 20),  # many symbols that occur once
        (np.concatenate([np.arange(256, dtype=np.uint8), workloads.runs(100000, 4)]), 1 << 20),
        (rs.randint(0, 256, size=5).astype(np.uint8), 16), (rs.randint(0, 256, size=15).astype(np.uint8), 16),
        (np.frombuffer(workloads.canterbury_files()["ptt5"], np.uint8), 65536),
    ]
    for data, block in cases:
        for coder, _ in CODERS:
            slots, sizes = oracle.encode_blocks(data, block, coder=coder, threads=8)
            payload, offsets, _ = gpu_encode(ctx, data, block, coder=coder)
            assert_same_blocks(payload, offsets, slots, sizes)
            back, st, _ = gpu_decode(ctx, payload, offsets, len(data), block, coder=coder)
            assert st == 0 and np.array_equal(back, data)


def test_rans_damaged_streams_are_reported(ctx, oracle):
    from cpprcoder_amd import rcx
    data = workloads.zipf(65536 * 3 + 500, 8)
    for coder, _ in CODERS:
        payload, offsets, _ = gpu_encode(ctx, data, 65536, coder=coder)
        cut = offsets.copy()
        cut[-1] -= 30  # the last block loses its tail
        _, st, blk = gpu_decode(ctx, payload[: int(cut[-1])], cut, len(data), 65536, coder=coder)
        assert st == rcx.E_CORRUPT and blk == 3
        bad = payload.copy()
        at = int(offsets[1]) + 4 + 4 * 200  # block 1: cum[200] above its successor
        bad[at: at + 4] = np.frombuffer(np.uint32(5000 if coder == 3 else 17000).tobytes(), np.uint8)
        _, st, blk = gpu_decode(ctx, bad, offsets, len(data), 65536, coder=coder)
        assert st == rcx.E_CORRUPT and blk == 1
        bad = payload.copy()
        bad[int(offsets[2])] ^= 1  # block 2 declares another size
        _, st, blk = gpu_decode(ctx, bad, offsets, len(data), 65536, coder=coder)
        assert st == rcx.E_CORRUPT and blk == 2
        back, st, _ = gpu_decode(ctx, payload, offsets, len(data), 65536, coder=coder)
        assert st == 0 and np.array_equal(back, data)
        # streams made by the CPU coder decode on the GPU
        slots, sizes = oracle.encode_blocks(data, 65536, coder=coder, threads=8)
        p2, o2 = oracle.compact(slots, sizes)
        back, st, _ = gpu_decode(ctx, p2, o2, len(data), 65536, coder=coder)
        assert st == 0 and np.array_equal(back, data)


def test_rans_full_size_round_trip(ctx, oracle):
    """256 MiB of Zipf bytes in 64 KiB blocks: round trip, and a sample of blocks byte for byte against the oracle."""
    from cpprcoder_amd import rcx
    n, block = 1 << 28, 65536
    nblocks = n // block
    data = workloads.zipf(n, 12345)
    src = torch.from_numpy(data).cuda()
    for coder, _ in CODERS:
        dst = torch.empty(rcx.encode_bound(n, block, coder), dtype=torch.uint8, device="cuda")
        offs = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
        ctx.encode_blocks_device(src, block, dst, offs, coder=coder)
        ctx.sync_status()
        offsets = offs.cpu().numpy()
        total = int(offsets[-1])
        assert 0.78 < total / n < 0.82
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctx.decode_blocks_device(dst, total, offs, n, block, out, coder=coder)
        ctx.sync_status()
        assert torch.equal(out, src)
        rs = np.random.RandomState(coder)
        for b in sorted(set([0, 1, 7, 8, nblocks - 1] + [int(x) for x in rs.randint(0, nblocks, 40)])):
            ref = oracle.rans_encode(data[b * block:(b + 1) * block], simd=(coder == 3))
            assert dst[int(offsets[b]): int(offsets[b + 1])].cpu().numpy().tobytes() == ref, f"block {b}"
