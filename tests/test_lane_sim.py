"""The device per-lane coder (cpprcoder_amd/csrc/rcx_lane.hpp) compiled for the host and
checked against the oracle, so the arithmetic is known good before it meets a GPU.

tests/sim/lane_sim.cpp is test tooling (g++, no HIP); it is not part of librcx.so.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cpprcoder_amd import workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_SRC = os.path.join(ROOT, "tests", "sim", "lane_sim.cpp")
SIM_SO = os.path.join(ROOT, "tests", "sim", "liblanesim.so")


@pytest.fixture(scope="module")
def sim():
    deps = [SIM_SRC] + [os.path.join(ROOT, "cpprcoder_amd", "csrc", f) for f in ("rcx_lane.hpp", "rcx_divtab.hpp", "rcx_bwt_tie.hpp")]
    if not os.path.exists(SIM_SO) or any(os.path.getmtime(d) > os.path.getmtime(SIM_SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SIM_SO, SIM_SRC], check=True)
    L = C.CDLL(SIM_SO)
    L.sim_check_divtab.restype, L.sim_check_divtab.argtypes = C.c_uint32, [C.c_uint32, C.c_uint32]
    L.sim_encode_blocks.restype = C.c_int
    L.sim_encode_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
    L.sim_decode_blocks.restype = C.c_uint64
    L.sim_decode_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32]
    L.sim_stream_encode_track.restype = None
    L.sim_stream_encode_track.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    L.sim_stream_decode_track.restype = C.c_uint32
    L.sim_stream_decode_track.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.sim_counters.restype, L.sim_counters.argtypes = None, [C.c_void_p, C.c_int]
    L.sim_stream_encode_long.restype, L.sim_stream_encode_long.argtypes = C.c_uint32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.sim_stream_decode_long.restype, L.sim_stream_decode_long.argtypes = C.c_uint32, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    return L


def counters(sim, reset=True):
    a = (C.c_uint64 * 4)()
    sim.sim_counters(a, int(reset))
    return list(a)


def test_divisor_table_is_exact(sim):
    # every total a block of up to RCX_MAX_BLOCK = 2^24 - 256 symbols can see, probed at the edges of each quotient
    assert sim.sim_check_divtab(256, 256 + (1 << 20) + 128) == 0
    for first in range(1 << 21, 1 << 24, 1 << 21):  # and samples of the rest up to 2^24 (+ the table's padding)
        assert sim.sim_check_divtab(first - 2000, first + 2000) == 0
    assert sim.sim_check_divtab((1 << 24) - 70000, (1 << 24) + 300) == 0


def roundtrip(sim, oracle, data, block, lane):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = len(data)
    slots, sizes = oracle.encode_blocks(data, block, threads=4)
    s2, z2 = np.zeros_like(slots), np.zeros_like(sizes)
    assert sim.sim_encode_blocks(data.ctypes.data, n, block, s2.ctypes.data, slots.shape[1], z2.ctypes.data, lane) == 0
    assert np.array_equal(sizes, z2)
    for b in range(len(sizes)):
        assert np.array_equal(slots[b, : sizes[b]], s2[b, : sizes[b]]), f"block {b}"
    payload, offs = oracle.compact(slots, sizes)
    out = np.zeros(n, dtype=np.uint8)
    assert sim.sim_decode_blocks(payload.ctypes.data, offs.ctypes.data, len(sizes), block, n, out.ctypes.data, lane) == 0
    assert np.array_equal(out, data)


@pytest.mark.parametrize("wl", ["uniform", "zipf", "canterbury", "runs"])
def test_lane_matches_oracle(sim, oracle, wl):
    for block, n, lane in ((4096, 1 << 19, 0), (65536, (1 << 21) - 777, 17), (262144, 1 << 21, 63), (16, 3000, 5), (48, 1000, 40)):
        roundtrip(sim, oracle, workloads.by_name(wl, n, 12345), block, lane)


def test_lane_edge_inputs(sim, oracle):
    counters(sim)
    roundtrip(sim, oracle, np.full(300000, 255, np.uint8), 65536, 1)
    roundtrip(sim, oracle, np.zeros(70000, np.uint8), 65536, 2)
    roundtrip(sim, oracle, np.full(1 << 20, 65, np.uint8), 1 << 20, 3)  # RCX_MAX_BLOCK, one symbol: f reaches 2^20
    roundtrip(sim, oracle, workloads.runs(1 << 21, 3), 1 << 20, 4)
    roundtrip(sim, oracle, np.arange(256, dtype=np.uint8).repeat(3), 768, 6)
    files = workloads.canterbury_files()
    roundtrip(sim, oracle, np.frombuffer(files["ptt5"], np.uint8), 65536, 7)        # long pending-0xFF runs
    roundtrip(sim, oracle, np.frombuffer(files["kennedy.xls"], np.uint8), 1 << 20, 8)
    slow_carries, bytes_touched, _, _ = counters(sim)
    assert slow_carries > 0 and bytes_touched >= slow_carries  # the carry-into-memory path was exercised


def test_stream_tracking_matches_reference_sink_semantics(sim, oracle):
    # rcx_stream_encode's second pass: where does a bounded MemoryStream fill? (cpprcoder.h:708-711, :744-755)
    rs = np.random.RandomState(5)
    cases = [(workloads.uniform(65536, 12345), 65536)]  # {Pending, 112}
    for _ in range(60):
        n = int(rs.randint(1, 5000))
        data = workloads.by_name(("uniform", "zipf", "runs")[int(rs.randint(3))], n, int(rs.randint(1 << 30)))
        full = oracle.adaptive_encode(data)[2]
        cases.append((data, int(rs.randint(16, full + 32))))
        cases.append((data, max(16, full - int(rs.randint(0, 12)))))
    for data, cap in cases:
        data = np.ascontiguousarray(data)
        (st, rq), ref_bytes, ref_size = oracle.adaptive_encode(data, sink_capacity=cap)
        cap16 = 16 if cap <= 0 else (cap + 15) & ~15
        slot = np.zeros(len(data) + len(data) // 32 + 1040, np.uint8)
        out = np.zeros(3, np.uint32)
        sim.sim_stream_encode_track(data.ctypes.data, len(data), cap16, slot.ctypes.data, len(slot), out.ctypes.data)
        fail_at, flush_fails, size = int(out[0]), int(out[1]), int(out[2])
        if size - 4 <= cap16:
            assert (st, rq) == (0, 0) and ref_size == size and bytes(slot[:size])[:len(ref_bytes)] == ref_bytes  # helper clips to cap
            assert fail_at == 0xFFFFFFFF and not flush_fails
        elif fail_at != 0xFFFFFFFF:
            assert (st, rq) == (1, len(data) - fail_at) and ref_size == cap16
            assert bytes(slot[:min(cap16, cap)]) == ref_bytes
        else:
            assert flush_fails and (st, rq) == (0, 0) and ref_size == cap16
            assert bytes(slot[:min(cap16, cap)]) == ref_bytes


def test_stream_decode_tracking(sim, oracle, golden):
    u = workloads.uniform(65536, 12345)[:3000]
    comp = oracle.adaptive_encode(u)[1]
    for cut in (1500, 8, 9, 100, len(comp) - 5, len(comp)):
        piece = np.frombuffer(comp[:cut], np.uint8).copy()
        (st, rq), ref_out, ref_size = oracle.adaptive_decode(piece, 3000)
        out = np.zeros(3000, np.uint8)
        short_at = sim.sim_stream_decode_track(piece.ctypes.data, len(piece), 3000, out.ctypes.data)
        if st == 1:
            assert short_at == ref_size and rq == 3000 - short_at and bytes(out[:short_at]) == ref_out
        else:
            assert short_at == 0xFFFFFFFF and bytes(out) == ref_out
    # corrupt input: the off-table fall-through of find() (cpprcoder.h:1220-1242)
    counters(sim)
    for key in ("decode_junk_64", "decode_allff_64"):
        pin = golden["kat"]["pins"][key]
        junk = np.frombuffer(bytes.fromhex(pin["input_hex"]), np.uint8).copy()
        out = np.zeros(64, np.uint8)
        short_at = sim.sim_stream_decode_track(junk.ctypes.data, len(junk), 64, out.ctypes.data)
        (st, rq), ref_out, _ = oracle.adaptive_decode(junk, 64)
        assert st == 0 and short_at == 0xFFFFFFFF and bytes(out) == ref_out == bytes.fromhex(pin["out_hex"])
    assert counters(sim)[2] > 0  # the off-table path was taken


def test_long_stream_through_the_halving(sim, oracle):
    # cpprcoder.h:1138-1176: past 2^24 - 256 symbols the table is halved; the lane's step_long path must produce the
    # oracle's bytes (the oracle's model is pinned through the halving by the reference's model probes)
    cases = [workloads.uniform((1 << 24) - 256 + 5000, 11),
             np.minimum(workloads.zipf((1 << 24) + 70000, 5), 3).astype(np.uint8),
             workloads.zipf(300000, 8)]  # the long path on a short stream
    for data in cases:
        n = len(data)
        (st, rq), ref_bytes, ref_size = oracle.adaptive_encode(data)
        assert st == 0
        slot = np.zeros(oracle.default_capacity(n) + 64, np.uint8)
        size = sim.sim_stream_encode_long(data.ctypes.data, n, slot.ctypes.data, len(slot) - 32)
        assert size == ref_size and bytes(slot[:size]) == ref_bytes
        out = np.zeros(n, np.uint8)
        comp = np.frombuffer(ref_bytes, np.uint8).copy()
        assert sim.sim_stream_decode_long(comp.ctypes.data, len(comp), n, out.ctypes.data) == 0xFFFFFFFF
        assert np.array_equal(out, data)


def test_tie_row_replay_matches_the_reference_rows(sim):
    """Periodic blocks: the replay of the reference's sort on row classes (rcx_bwt_tie.hpp) lands on the row the real
    reference stored (tests/golden/bwt.json)."""
    import json

    import bwt_cases
    with open(os.path.join(ROOT, "tests", "golden", "bwt.json")) as f:
        want = json.load(f)["cases"]
    sim.sim_bwt_tie_row.restype = C.c_uint32
    sim.sim_bwt_tie_row.argtypes = [C.c_void_p, C.c_uint32]
    seen = 0
    for name, data in bwt_cases.cases().items():
        if len(data) != bwt_cases.BLOCK:
            continue
        p = bwt_cases.primitive_period(data)
        if p == bwt_cases.BLOCK:
            continue
        block = np.ascontiguousarray(data)
        assert sim.sim_bwt_tie_row(block.ctypes.data, p) == want[name]["rows"][0], name
        seen += 1
    assert seen >= 10


def test_wave_wide_tie_replay_moves_every_row_like_the_scalar_one(sim):
    """csrc/rcx_bwt_tie.hpp places runs of equal rows, passes rows that only move a pointer and exchanges the equal zones
    64 rows at a time; the reference makes those moves one by one.  Every row must end where the one-at-a-time replay (the
    round-2 code, kept in tests/sim/lane_sim.cpp) leaves it: words of few symbols (long equal runs behind long queues),
    of many (queues of every length), runs with single odd bytes, sorted and skewed words, depths 256 ... 32768."""
    sim.sim_bwt_tie_compare.restype = C.c_uint32
    sim.sim_bwt_tie_compare.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    rs = np.random.RandomState(7)
    words = []
    for p in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        for alphabet in (2, 3, 5, 17, 256):
            words.append((rs.randint(0, alphabet, p).astype(np.uint8), p))
        run = np.full(p, 97, np.uint8)
        run[-1] = 98
        words.append((run.copy(), p))              # a^(p-1) b
        run[0] = 96
        words.append((run.copy(), p))              # ` a^(p-2) b
        words.append((np.sort(rs.randint(0, 256, p)).astype(np.uint8), p))
        words.append((np.minimum(rs.randint(0, 256, p), 9).astype(np.uint8), p))
    checked = 0
    for word, p in words:
        if p > 1 and all(np.array_equal(word, np.roll(word, -q)) for q in (p // 2,)):
            continue  # (not primitive: such a block has a smaller period and is listed with that)
        for depth in sorted({max(2 * p, 256), 8 * p if 8 * p <= 32768 else 32768, 32768 if p >= 1024 else 4096}):
            if depth < 2 * p:
                continue
            w = np.ascontiguousarray(word)
            assert sim.sim_bwt_tie_compare(w.ctypes.data, p, depth) == 0, (p, depth, word[:8])
            checked += 1
    assert checked >= 150
