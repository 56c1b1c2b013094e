"""GPU parity of the block sort (rcx_bwt_*; blksort.h:426-679) through the C ABI: the encoded bytes -- last column AND
the stored row index, periodic blocks included -- must be what the real reference wrote (tests/golden/bwt.json, made
from the reference build by tests/golden/make_golden_bwt.py), and decode must give what BlkSort::decode gives, for
transformed blocks and for arbitrary bytes (the oracle restates blksort.h and is itself pinned to that build)."""
import hashlib
import json
import os

import numpy as np
import pytest

import bwt_cases
import oracle_lib
from cpprcoder_amd import workloads

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK, ENCODED = bwt_cases.BLOCK, bwt_cases.ENCODED


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module", params=["ballot", "atomic"])
def ctx(request):
    """Every test of this file runs in both forms of the counting pass's rank (csrc/rcx_bwt.hpp): with ballots -- the
    default, documented behaviour only -- and with one returning LDS atomic per key, which a caller opts into with
    RCX_BWT_MATCH=atomic (read at the context's first block-sort call, so it stays set while the context lives)."""
    from cpprcoder_amd import rcx
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    before = os.environ.get("RCX_BWT_MATCH")
    os.environ["RCX_BWT_MATCH"] = request.param
    c = rcx.Context(0)
    yield c
    c.close()
    if before is None:
        os.environ.pop("RCX_BWT_MATCH", None)
    else:
        os.environ["RCX_BWT_MATCH"] = before


@pytest.fixture(scope="module")
def golden_bwt():
    with open(os.path.join(HERE, "golden", "bwt.json")) as f:
        return json.load(f)


def test_bounds():
    from cpprcoder_amd import rcx
    L = rcx.lib()
    for n in (0, 1, 32767, 32768, 32769, 65536, 100000, 3 * 32768 + 5, (1 << 32) + 12345):
        assert L.rcx_bwt_encode_bound(n) == n + 2 * (n // 32768)          # blksort.h:426-431
        assert L.rcx_bwt_decode_bound(n) == n                             # blksort.h:433-438
        assert L.rcx_bwt_decoded_size(L.rcx_bwt_encode_bound(n)) == n     # blksort.h:451-462


def test_every_fixture_case_is_the_reference_output(ctx, golden_bwt):
    for name, data in bwt_cases.cases().items():
        want = golden_bwt["cases"][name]
        enc = ctx.bwt_encode(data)
        assert len(enc) == want["encoded_size"], name
        assert bwt_cases.rows(enc) == want["rows"], name            # the row index, ties included
        assert sha(enc) == want["encoded_sha256"], name
        back = ctx.bwt_decode(enc)
        assert np.array_equal(back, data), name


def test_tie_replay_runs_only_for_periodic_blocks(ctx):
    c = bwt_cases.cases()
    ctx.bwt_encode(c["random 2 blocks + tail"])
    assert ctx.bwt_last_ties() == 0
    ctx.bwt_encode(c["zeros"])                                     # period 1: the row is 0 without any replay
    assert ctx.bwt_last_ties() == 0
    ctx.bwt_encode(c["ones then a two"])                           # almost periodic is not periodic
    assert ctx.bwt_last_ties() == 0
    ctx.bwt_encode(c["three blocks: periodic, random, zeros"])
    assert ctx.bwt_last_ties() == 1
    many = np.concatenate([bwt_cases.periodic(2 << (i % 13), 300 + i) for i in range(40)])
    enc = ctx.bwt_encode(many)
    assert ctx.bwt_last_ties() == 40
    assert np.array_equal(enc, oracle_lib.oracle().bwt_encode(many, threads=16))


def test_harness_files(ctx, golden_bwt):
    """The files run_blksort is run over (test/main.cpp:1248-1258, :1327-1333)."""
    files = workloads.canterbury_files()
    for name in workloads.CANTERBURY_ORDER:
        data = np.frombuffer(files[name], np.uint8)
        want = golden_bwt["canterbury"][name]
        enc = ctx.bwt_encode(data)
        assert (len(enc), sha(enc)) == (want["encoded_size"], want["encoded_sha256"]), name
        assert bwt_cases.rows(enc) == want["rows"], name
        assert np.array_equal(ctx.bwt_decode(enc), data), name


def test_fresh_data_against_the_oracle(ctx):
    o = oracle_lib.oracle()
    for wl, seed in (("uniform", 5), ("zipf", 6), ("canterbury", 7), ("runs", 8)):
        data = workloads.by_name(wl, 24 * BLOCK + 4321, seed)
        enc = ctx.bwt_encode(data)
        assert np.array_equal(enc, o.bwt_encode(data, threads=16)), wl
        assert np.array_equal(ctx.bwt_decode(enc), data), wl


def test_many_shapes_against_the_oracle(ctx):
    """Blocks of many shapes in one buffer: small alphabets, runs, repeats of odd lengths, sorted bytes, and periodic
    blocks with a few bytes changed (almost all rotations tie over thousands of bytes, none over all of them)."""
    o = oracle_lib.oracle()
    mix, blocks = bwt_cases.mix, []
    for i in range(40):
        a = 2 + (i * 37) % 255
        blocks.append((mix(BLOCK, 2000 + i).astype(np.uint32) % a).astype(np.uint8))
    for i in range(16):
        lens = 1 + (mix(BLOCK, 2100 + i).astype(np.uint32) % (3 + 40 * i))
        vals = mix(BLOCK, 2200 + i)
        blocks.append(np.repeat(vals, lens)[:BLOCK])
    for i, period in enumerate((3, 5, 7, 100, 1000, 4097, 10000, 16383, 21845)):
        blocks.append(np.resize(mix(period, 2300 + i), BLOCK).astype(np.uint8))
    blocks.append(np.sort(mix(BLOCK, 2400)))
    blocks.append(np.sort(mix(BLOCK, 2401))[::-1].copy())
    for i, (p, changes) in enumerate(((1, 1), (2, 1), (4, 2), (16, 1), (256, 3), (1024, 1), (4096, 5), (16384, 1))):
        b = bwt_cases.periodic(p, 2500 + i, 256 if i % 2 else 3).copy()
        for c in range(changes):
            b[(7919 * (c + 1) * (i + 3)) % BLOCK] ^= 0x5A
        blocks.append(b)
    data = np.concatenate(blocks)
    enc = ctx.bwt_encode(data)
    assert ctx.bwt_last_ties() == 0
    want = o.bwt_encode(data, threads=16)
    for b in range(len(blocks)):
        assert np.array_equal(enc[b * ENCODED:(b + 1) * ENCODED], want[b * ENCODED:(b + 1) * ENCODED]), b
    assert np.array_equal(ctx.bwt_decode(enc), data)


def test_inverse_of_arbitrary_bytes_is_the_reference_walk(ctx):
    """BlkSort::decode is defined for any column and any row below 32768 (the walk just follows a permutation, possibly
    round a short cycle): the GPU must follow it the same way."""
    o = oracle_lib.oracle()
    rng_blocks = []
    for i in range(12):
        col = bwt_cases.mix(BLOCK, 900 + i) if i % 3 else (bwt_cases.mix(BLOCK, 900 + i) % 3).astype(np.uint8)
        row = (i * 2731 + 17) % BLOCK
        rng_blocks.append(np.concatenate([col, np.array([row & 0xFF, row >> 8], np.uint8)]))
    # a periodic block's column with every row of one tie, and a constant column (identity permutation)
    per = o.bwt_encode(bwt_cases.periodic(8, 21))
    for row in (0, 1, 4095, 4096, 20000, 32767):
        e = per.copy()
        e[BLOCK], e[BLOCK + 1] = row & 0xFF, row >> 8
        rng_blocks.append(e)
    rng_blocks.append(np.concatenate([np.full(BLOCK, 7, np.uint8), np.array([0x34, 0x12], np.uint8)]))
    enc = np.concatenate(rng_blocks + [bwt_cases.mix(999, 77)])
    assert np.array_equal(ctx.bwt_decode(enc), o.bwt_decode(enc, threads=8))


def test_row_past_the_block_is_refused(ctx):
    from cpprcoder_amd import rcx
    enc = ctx.bwt_encode(bwt_cases.mix(2 * BLOCK, 31)).copy()
    enc[ENCODED + BLOCK + 1] |= 0x80                                 # block 1: row >= 32768 (blksort.h:663 would read past its array)
    with pytest.raises(rcx.RcxError) as e:
        ctx.bwt_decode(enc)
    assert e.value.status == rcx.E_CORRUPT
    assert np.array_equal(ctx.bwt_decode(ctx.bwt_encode(b"still works")), np.frombuffer(b"still works", np.uint8))


def test_device_calls_any_alignment_and_capacity(ctx):
    from cpprcoder_amd import rcx
    o = oracle_lib.oracle()
    # text, and blocks of long runs (which start from run keys: another path through the block in LDS), and a tail
    data = np.concatenate([workloads.by_name("canterbury", 3 * BLOCK, 3), workloads.by_name("runs", 2 * BLOCK + 100, 4)])
    want = o.bwt_encode(data, threads=8)
    for off_in, off_out in ((0, 0), (1, 0), (0, 3), (7, 9), (15, 1)):
        src = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
        src[off_in: off_in + len(data)] = torch.from_numpy(data.copy()).cuda()
        dst = torch.full((len(want) + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        ctx.bwt_encode_device(src[off_in: off_in + len(data)], dst[off_out: off_out + len(want)])
        ctx.sync_status()
        got = dst.cpu().numpy()
        assert np.array_equal(got[off_out: off_out + len(want)], want), (off_in, off_out)
        assert (got[:off_out] == 0xA5).all() and (got[off_out + len(want):] == 0xA5).all()
        back = torch.full((len(data) + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        ctx.bwt_decode_device(dst[off_out: off_out + len(want)], len(want), back[off_in: off_in + len(data)])
        ctx.sync_status()
        b = back.cpu().numpy()
        assert np.array_equal(b[off_in: off_in + len(data)], data), (off_in, off_out)
        assert (b[:off_in] == 0x5A).all() and (b[off_in + len(data):] == 0x5A).all()
    src = torch.zeros(2 * BLOCK, dtype=torch.uint8, device="cuda")
    small = torch.zeros(2 * ENCODED - 1, dtype=torch.uint8, device="cuda")
    with pytest.raises(rcx.RcxError) as e:
        ctx.bwt_encode_device(src, small)
    assert e.value.status == rcx.E_CAPACITY


def test_a_gigabyte_round_trip_property(ctx):
    """BASELINE-sized input: no oracle at this size; decode(encode(x)) == x and the structure of the output
    (blksort.h:440-449): every block's column is a permutation of the block."""
    n = 1 << 30
    data = torch.from_numpy(workloads.by_name("zipf", 1 << 26, 12345)).cuda().repeat(n >> 26)
    # make the copies differ so that blocks are not repeats of each other
    data[:: 4099] ^= 0x55
    enc = torch.empty(rcx_bound(n), dtype=torch.uint8, device="cuda")
    ctx.bwt_encode_device(data, enc)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.bwt_decode_device(enc, enc.numel(), back)
    ctx.sync_status()
    assert torch.equal(back, data)
    cols = enc.view(-1, ENCODED)[:, :BLOCK]
    assert int(cols.to(torch.int64).sum()) == int(data.to(torch.int64).sum())
    assert ctx.bwt_last_ties() == 0


def rcx_bound(n: int) -> int:
    from cpprcoder_amd import rcx
    return rcx.bwt_encode_bound(n)
