"""The CPU restatement (oracle/) against the golden vectors made from the real reference,
and against the real reference build itself when oracle/_ref/ is present.

Mirrors what the reference's own harness checks (round trip, test/main.cpp:357-361) plus the
only pinned results the reference publishes (README.md:16-46 ratios).
"""
import hashlib

import numpy as np
import pytest

import oracle_lib
from cpprcoder_amd import workloads


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


GENERATED = {
    "A*65535": lambda: b"A" * 65535, "A*65536": lambda: b"A" * 65536, "A*65537": lambda: b"A" * 65537,
    "ff*70000": lambda: b"\xff" * 70000,
    "uniform(65536,12345)": lambda: workloads.uniform(65536, 12345).tobytes(),
    "uniform(100000,7)": lambda: workloads.uniform(100000, 7).tobytes(),
    "zipf(65536,12345)": lambda: workloads.zipf(65536, 12345).tobytes(),
    "runs(200000,7)": lambda: workloads.runs(200000, 7).tobytes(),
}


def test_small_kats(oracle, golden):
    for k in golden["kat"]["kat"]:
        v = bytes.fromhex(k["input_hex"])
        st, out, size = oracle.adaptive_encode(v)
        assert out.hex() == k["adaptive_hex"] and list(st) == k["adaptive_status"]
        ok, sout, ssize = oracle.static_encode(v)
        assert ok == k["static_ok"] and ssize == k["static_size"] and sha(sout) == k["static_sha256"]
        assert sout[516:].hex() == k["static_tail_hex"]
        # round trip (the decoder emits one byte for the empty stream, cpprcoder.h:912)
        st, back, _ = oracle.adaptive_decode(out, max(len(v), 16))
        assert st == (0, 0) and back == (v if v else b"\x00")
        ok, back, _ = oracle.static_decode(sout, max(len(v), 16))
        assert ok and back == v


def test_survey_vectors(oracle):
    # streams captured from the reference during the survey
    table = {"": "000000000000000000", "00": "01000000000000000000", "ff": "0100000000feffff0100",
             "6162": "020000000061629d013a00", b"abcabcabcabc".hex(): "0c00000000616301347bf00c7159aaba82e400",
             b"hello world".hex(): "0b00000000686508218531aa29577576ba5c00",
             "ff" * 16: "1000000000fffffeff1bb2aeedb6dd63d1517b", "00" * 16: "10000000" + "00" * 15}
    for i, o in table.items():
        assert oracle.adaptive_encode(bytes.fromhex(i))[1].hex() == o
    assert oracle.adaptive_encode(b"A" * 65536)[2] == 311
    assert oracle.adaptive_encode(bytes(range(256)))[2] == 282
    assert oracle.static_encode(b"hello world")[1][-8:].hex() == "004b21a69d0a028f"
    for n in (65535, 65536, 65537):
        assert oracle.static_encode(b"A" * n)[2] == 521
    assert oracle.static_encode(bytes(range(256)))[2] == 777


def test_generated(oracle, golden):
    for name, g in golden["kat"]["generated"].items():
        v = GENERATED[name]()
        assert len(v) == g["n"] and sha(v) == g["input_sha256"], name
        st, out, size = oracle.adaptive_encode(v)
        assert (size, sha(out), list(st)) == (g["adaptive_size"], g["adaptive_sha256"], g["adaptive_status"]), name
        assert "%016x" % oracle_lib.fnv1a64(out) == g["adaptive_fnv1a64"]
        ok, sout, ssize = oracle.static_encode(v)
        assert (ok, ssize, sha(sout)) == (g["static_ok"], g["static_size"], g["static_sha256"]), name
        assert oracle.adaptive_decode(out, len(v))[1] == v
        assert oracle.static_decode(sout, len(v))[1] == v


# README.md:20-30 (static) and :36-46 (adaptive): ratio = compressed / original, 6 decimals
README_RATIOS = {
    "alice29.txt": (0.574532, 0.573000), "asyoulik.txt": (0.605293, 0.603400), "cp.html": (0.674836, 0.662480),
    "fields.c": (0.672646, 0.642511), "grammar.lsp": (0.718893, 0.619457), "kennedy.xls": (0.452938, 0.447426),
    "lcet10.txt": (0.585129, 0.584625), "plrabn12.txt": (0.567788, 0.567367), "ptt5": (0.157010, 0.152158),
    "sum": (0.679759, 0.670450), "xargs.1": (0.735510, 0.648924),
}


def test_canterbury_matches_readme_and_golden(oracle, golden):
    files = workloads.canterbury_files()
    for name in workloads.CANTERBURY_ORDER:
        v, g = files[name], golden["kat"]["canterbury"][name]
        assert len(v) == g["n"]
        st, out, size = oracle.adaptive_encode(v)
        ok, sout, ssize = oracle.static_encode(v)
        assert (size, sha(out)) == (g["adaptive_size"], g["adaptive_sha256"]), name
        assert (ssize, sha(sout)) == (g["static_size"], g["static_sha256"]), name
        s_ratio, a_ratio = README_RATIOS[name]
        assert round(size / len(v), 6) == pytest.approx(a_ratio, abs=1.1e-6), name
        assert round(ssize / len(v), 6) == pytest.approx(s_ratio, abs=1.1e-6), name
        assert oracle.adaptive_decode(out, len(v))[1] == v
        assert oracle.static_decode(sout, len(v))[1] == v


def test_block_tables(oracle, golden):
    for t in golden["blocks"]:
        if t["block"] * len(t["sizes"]) > (6 << 20):
            continue  # keep the CPU suite short; the GPU suite covers every table
        data = workloads.by_name(t["workload"], t["n"], t["seed"])
        assert sha(data) == t["input_sha256"]
        coder = 0 if t["coder"] == "adaptive" else 1
        slots, sizes = oracle.encode_blocks(data, t["block"], coder=coder, threads=4)
        assert [int(s) for s in sizes] == t["sizes"], (t["workload"], t["block"])
        assert ["%016x" % oracle_lib.fnv1a64(slots[b, : int(sizes[b])]) for b in range(len(sizes))] == t["fnv1a64"]
        back, ok = oracle.decode_blocks(slots, sizes, t["block"], t["n"], coder=coder, threads=4)
        assert ok and np.array_equal(back, data)


def test_pins(oracle, golden):
    p = golden["kat"]["pins"]
    u64k = workloads.uniform(65536, 12345)
    st, out, size = oracle.adaptive_encode(u64k, sink_capacity=65536)  # {Pending,112}
    g = p["overflow_uniform64k_into_65536"]
    assert (list(st), size, sha(out)) == (g["status"], g["size"], g["prefix_sha256"]) and list(st) == [1, 112]
    st, out, _ = oracle.adaptive_decode(bytes.fromhex("000000000000000000"), 16)
    assert (list(st), out.hex()) == (p["decode_of_empty_stream"]["status"], p["decode_of_empty_stream"]["out_hex"])
    st, out, _ = oracle.adaptive_decode(b"\x01\x00\x00", 16)
    assert list(st) == p["decode_short_input"]["status"] == [1, 8]
    for key in ("decode_junk_64", "decode_allff_64"):  # corrupt input incl. the target >= total fall-through
        st, out, _ = oracle.adaptive_decode(bytes.fromhex(p[key]["input_hex"]), 64)
        assert (list(st), out.hex()) == (p[key]["status"], p[key]["out_hex"]), key
    comp = oracle.adaptive_encode(u64k[:3000])[1]
    st, out, size = oracle.adaptive_decode(comp[:1500], 3000)
    assert (list(st), size, sha(out)) == (p["decode_truncated"]["status"], p["decode_truncated"]["size"], p["decode_truncated"]["out_sha256"])
    st, out, size = oracle.adaptive_decode(comp, 1000)
    assert (list(st), size, sha(out)) == (p["decode_sink_full"]["status"], p["decode_sink_full"]["size"], p["decode_sink_full"]["out_sha256"])


def test_sink_scripts(oracle, golden):
    for s in golden["kat"]["pins"]["stream_scripts"]:
        assert [list(t) for t in oracle.stream_script([tuple(o) for o in s["ops"]])] == s["trace"]


def test_model_probes(oracle, golden):
    gens = {"none": lambda: np.zeros(0, np.uint8), "zipf50k": lambda: workloads.zipf(50000, 3),
            "uniform_to_halving": lambda: workloads.uniform((1 << 24) - 256 + 5000, 11),
            "skew_to_halving": lambda: np.minimum(workloads.zipf((1 << 24) - 256, 5), 3).astype(np.uint8)}
    for g in golden["kat"]["pins"]["model_probes"]:
        syms = gens[g["label"]]()
        assert sha(syms) == g["input_sha256"]
        total, freq, cum, fcount, fcode = oracle.model_probe(syms, g["targets"])
        assert total == g["total"] and sha(freq.tobytes()) == g["freq_sha256"] and sha(cum.tobytes()) == g["cum_sha256"]
        assert [int(x) for x in fcount] == g["found_count"] and [int(x) for x in fcode] == g["found_code"]


def test_chunked_equals_oneshot(oracle):
    # SURVEY.md section 8(c) property tests: any split, byte-wise encode(u8), chunked decode
    for v in (workloads.zipf(20000, 5).tobytes(), workloads.runs(30000, 2).tobytes(), b"x"):
        one = oracle.adaptive_encode(v)
        for piece in (0, 1, 777, 4096, len(v)):
            assert oracle.adaptive_encode_chunked(v, piece) == one
        for piece in (1, 8, 1000, len(one[1])):
            st, back, _ = oracle.adaptive_decode_chunked(one[1], piece, len(v))
            assert st == (0, 0) and back == v


def test_against_reference_build(oracle, reference):
    if reference is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    rs = np.random.RandomState(2024)
    cases = [workloads.uniform(5000, 1), workloads.zipf(70000, 2), workloads.runs(150000, 3), workloads.canterbury_tiled(100000)[7:],
             np.zeros(1, np.uint8), np.full(300000, 255, np.uint8)]
    for _ in range(40):
        n = int(rs.randint(0, 3000))
        alpha = int(rs.randint(1, 257))
        cases.append(rs.randint(0, alpha, size=n).astype(np.uint8))
    for v in cases:
        assert oracle.adaptive_encode(v) == reference.adaptive_encode(v)
        assert oracle.static_encode(v) == reference.static_encode(v)
        comp = oracle.adaptive_encode(v)[1]
        assert oracle.adaptive_decode(comp, max(len(v), 1)) == reference.adaptive_decode(comp, max(len(v), 1))
        for cap in (len(comp) // 2, 16):
            assert oracle.adaptive_encode(v, sink_capacity=cap) == reference.adaptive_encode(v, sink_capacity=cap)
        for piece in (0, 333):
            assert oracle.adaptive_encode_chunked(v, piece) == reference.adaptive_encode_chunked(v, piece)
        # corrupt / random streams must decode to the same bytes and status too
        junk = rs.randint(0, 256, size=max(len(comp), 12)).astype(np.uint8)
        junk[:4] = np.frombuffer(np.uint32(min(len(v), 500)).tobytes(), np.uint8)
        assert oracle.adaptive_decode(junk, 600) == reference.adaptive_decode(junk, 600)
    data = workloads.zipf(1 << 20, 9)
    for block in (4096, 65536):
        so, zo = oracle.encode_blocks(data, block, threads=4)
        sr, zr = reference.encode_blocks(data, block, threads=4)
        assert np.array_equal(zo, zr) and np.array_equal(so, sr)


def long_inputs():
    """The recipes of tests/golden/make_golden_long.py (label -> bytes)."""
    no_halving = (1 << 24) - 256
    adaptive = {
        "uniform(2MiB+77,3)": lambda: workloads.uniform((2 << 20) + 77, 3),
        "zipf(NO_HALVING,4)": lambda: workloads.zipf(no_halving, 4),
        "uniform(NO_HALVING+5000,11)": lambda: workloads.uniform(no_halving + 5000, 11),
        "min(zipf(2^24+70000,5),3)": lambda: np.minimum(workloads.zipf((1 << 24) + 70000, 5), 3).astype(np.uint8),
    }
    static = {
        "zipf(2^24+1000,6)": lambda: workloads.zipf((1 << 24) + 1000, 6),
        "runs(3MiB,2)": lambda: workloads.runs(3 << 20, 2),
    }
    return adaptive, static


def test_long_streams_and_big_blocks(oracle, golden):
    # past 1 MiB, up to and through the halving at total = 2^24 (cpprcoder.h:1138-1176) and the static coder's
    # second rescale (cpprcoder.h:561-570): the restatement against the reference's own output
    adaptive, static = long_inputs()
    for label, g in golden["long"]["adaptive"].items():
        v = adaptive[label]()
        assert sha(v) == g["input_sha256"]
        (st, rq), comp, size = oracle.adaptive_encode(v)
        assert (st, rq, size, sha(comp)) == (0, 0, g["size"], g["sha256"]), label
        (st, rq), back, _ = oracle.adaptive_decode(comp, len(v))
        assert (st, rq) == (0, 0) and back == v.tobytes()
    for label, g in golden["long"]["static"].items():
        v = static[label]()
        assert sha(v) == g["input_sha256"]
        ok, comp, size = oracle.static_encode(v)
        assert ok and (size, sha(comp)) == (g["size"], g["sha256"]), label
    for t in golden["long"]["blocks"]:
        data = workloads.by_name(t["workload"], t["n"], t["seed"])
        assert sha(data) == t["input_sha256"]
        slots, sizes = oracle.encode_blocks(data, t["block"], coder=0 if t["coder"] == "adaptive" else 1, threads=4)
        assert [int(x) for x in sizes] == t["sizes"]
        assert ["%016x" % oracle_lib.fnv1a64(slots[b, : int(sizes[b])]) for b in range(len(sizes))] == t["fnv1a64"]


def test_encode_traces_from_the_reference_build(oracle, reference):
    """AdaptiveRangeEncoder<T>::encode in pieces: the sink's size after initialize() and after every call, the last call's
    result and the sink's bytes -- tests/golden/encode_traces.json, made by the real reference (make_golden_traces.py)."""
    import hashlib
    import json
    import os

    import trace_cases
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "encode_traces.json")) as f:
        want = {c["name"]: c for c in json.load(f)["cases"]}
    seen = 0
    for name, data, piece, cap in trace_cases.cases():
        w = want[name]
        assert hashlib.sha256(bytes(data)).hexdigest() == w["input_sha256"], name
        for chk in (oracle, reference):
            if chk is None:
                continue
            (st, rq), sink, sizes = chk.adaptive_encode_trace(data, piece, cap)
            assert [st, rq] == w["status"] and sizes == w["sink_sizes"] and hashlib.sha256(sink).hexdigest() == w["sink_sha256"], (name, chk.kind)
        seen += 1
    assert seen >= 15
