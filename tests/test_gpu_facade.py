"""The C++ facade (include/cpprcoder_amd/cpprcoder.h) compiled with g++ and run on the GPU box:
a caller written against the reference's class API gets the reference's bytes and return values
(cpprcoder.h:626-940), with the coding done on the MI355X."""
import os
import subprocess

import numpy as np
import pytest

from cpprcoder_amd import workloads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("facade") / "facade_test")
    lib = os.path.join(ROOT, "cpprcoder_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"),
                    "-o", out, "-L", lib, "-lrcx", f"-Wl,-rpath,{lib}"], check=True)
    return out


def run(exe, tmp_path, mode, data, *args):
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    src.write_bytes(bytes(data))
    p = subprocess.run([exe, mode, str(src), str(dst)] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stderr)
    return [int(x) for x in p.stdout.split()], dst.read_bytes()


def test_facade_encode_decode_like_the_reference(exe, tmp_path, oracle):
    cases = [b"", b"hello world", workloads.zipf(20000, 5).tobytes(), workloads.uniform(65536, 12345).tobytes(),
             workloads.canterbury_files()["alice29.txt"]]
    for v in cases:
        n = len(v)
        (rst, rrq), rout, rsize = oracle.adaptive_encode(v, sink_capacity=n + n // 32 + 1024)
        for piece in (0, 777, -1) if n <= 20000 else (0, 4096):
            (st, rq, size, cap), out = run(exe, tmp_path, "enc", v, n + n // 32 + 1024, piece)
            assert (st, rq, size) == (rst, rrq, rsize) and out == rout, (n, piece)
        for piece in (0, 1000):
            (st, rq, size, cap), back = run(exe, tmp_path, "dec", rout, max(n, 16), piece)
            assert (st, rq) == (0, 0) and back == (v if v else b"\x00")
    assert oracle.adaptive_encode(cases[-1])[2] == 87147  # README.md:36


def test_facade_full_sink(exe, tmp_path, oracle, golden):
    u = workloads.uniform(65536, 12345)
    (st, rq, size, cap), out = run(exe, tmp_path, "enc", u.tobytes(), 65536, 0)   # {Pending, 112}
    (rst, rrq), rout, rsize = oracle.adaptive_encode(u, sink_capacity=65536)
    assert (st, rq, size) == (rst, rrq, rsize) == (1, 112, 65536) and out == rout
    rs = np.random.RandomState(3)
    for _ in range(6):
        n = int(rs.randint(100, 3000))
        v = workloads.zipf(n, int(rs.randint(1 << 20))).tobytes()
        full = oracle.adaptive_encode(v)[2]
        cap = int(rs.randint(16, full))
        (rst, rrq), rout, rsize = oracle.adaptive_encode(v, sink_capacity=cap)
        (st, rq, size, _), out = run(exe, tmp_path, "enc", v, cap, 0)
        assert (st, rq, size) == (rst, rrq, rsize) and out[: len(rout)] == rout
    comp = oracle.adaptive_encode(u[:3000])[1]
    (st, rq, size, _), out = run(exe, tmp_path, "dec", comp[:1500], 3000, 0)      # input runs dry
    pin = golden["kat"]["pins"]["decode_truncated"]
    assert [st, rq] == pin["status"] and size == pin["size"]
    (st, rq, size, _), out = run(exe, tmp_path, "dec", comp, 1000, 0)             # sink fills
    pin = golden["kat"]["pins"]["decode_sink_full"]
    assert [st, rq] == pin["status"] and size == pin["size"]


def test_facade_block_coder(exe, tmp_path, oracle):
    data = workloads.canterbury_tiled(700_001)
    (total, noff), comp = run(exe, tmp_path, "blocks", data.tobytes(), 65536)
    slots, sizes = oracle.encode_blocks(data, 65536, threads=4)
    payload, offsets = oracle.compact(slots, sizes)
    assert total == len(payload) and noff == len(offsets) and comp == payload.tobytes()


def test_facade_static_range_encoder(exe, tmp_path, oracle):
    """RangeEncoder<>::encode / decode (cpprcoder.h:375-519) through the facade: same bytes, same bool."""
    files = workloads.canterbury_files()
    cases = [b"", b"hello world", bytes(range(256)), b"A" * 65536, workloads.zipf(30000, 2).tobytes(), files["alice29.txt"], files["sum"]]
    for v in cases:
        n = len(v)
        ok_ref, ref, size_ref = oracle.static_encode(v, sink_capacity=n + n // 32 + 2048)
        (ok, _, size, _), out = run(exe, tmp_path, "senc", v, n + n // 32 + 2048)
        assert (bool(ok), size) == (ok_ref, size_ref) and out == ref, n
        (ok, _, size, _), back = run(exe, tmp_path, "sdec", ref, max(n, 16))
        assert ok == 1 and back == v
    # a sink that fills: false, and the sink holds what had been written (cpprcoder.h:409-427)
    v = workloads.uniform(5000, 4).tobytes()
    full = oracle.static_encode(v)[1]
    for cap in (600, 1000, len(full) - 6):
        ok_ref, ref, size_ref = oracle.static_encode(v, sink_capacity=cap)
        (ok, _, size, _), out = run(exe, tmp_path, "senc", v, cap)
        assert (bool(ok), size) == (ok_ref, size_ref) and out[: len(ref)] == ref, cap
    # truncated input and a too-small sink on decode: false, with the symbols decoded so far
    for piece, cap in ((full[: len(full) // 2], 5000), (full[:520], 5000), (full[:516], 5000), (full[:100], 5000), (full, 1000)):
        ok_ref, ref, size_ref = oracle.static_decode(piece, cap)
        (ok, _, size, _), out = run(exe, tmp_path, "sdec", piece, cap)
        assert (bool(ok), size) == (ok_ref, size_ref) and out[: len(ref)] == ref, (len(piece), cap)


def test_facade_rans(exe, tmp_path, oracle):
    """cppans::rANS::encode / decode / encode_simd / decode_simd (cppans.h:497-649) through include/cpprcoder_amd/cppans.h:
    same bytes at the END of the destination, same return values (decode: payload bytes consumed, decode_simd: n)."""
    files = workloads.canterbury_files()
    cases = [b"hello world", bytes(range(256)), b"A" * 5000, workloads.zipf(30000, 2).tobytes(), files["fields.c"], b"xyz"]
    for v in cases:
        n = len(v)
        for simd in (0, 1):
            if simd and n < 16:
                continue  # the reference's own calc_encoded_size is too small for encode_simd below 16 symbols
            ref = oracle.rans_encode(v, bool(simd))
            (size, cap), out = run(exe, tmp_path, "renc", v, 0, simd)
            assert cap == 2 * n + 1032
            if len(ref) > cap:  # one repeated byte through encode_simd: 2n + 1064 bytes, more than calc_encoded_size allows
                assert size == 0  # cppans.h:599-601 (the reference has written in front of its destination by then)
                continue
            assert size == len(ref) and out == ref, (n, simd)
            (ret, _), back = run(exe, tmp_path, "rdec", ref, n, simd)
            assert back == v and ret == (n if simd else len(ref) - 1032), (n, simd)
            (ret, _), _ = run(exe, tmp_path, "rdec", ref, n - 1, simd)  # destination too small: 0 (cppans.h:541, :618)
            assert ret == 0
            small = 1100 if not simd else 1130
            (size, cap), out = run(exe, tmp_path, "renc", v, small, simd)  # a stream that does not fit: 0 (cppans.h:522, :599)
            assert size == (len(ref) if len(ref) <= small else 0)


def test_facade_blksort(exe, tmp_path, oracle):
    """blksort::BlkSort::encode / decode (blksort.h:440-462) through include/cpprcoder_amd/blksort.h, as run_blksort
    calls them (test/main.cpp:812-825)."""
    import bwt_cases
    files = workloads.canterbury_files()
    for v in (files["alice29.txt"], b"short", workloads.zipf(3 * 32768, 9).tobytes(), bwt_cases.periodic(64, 15, 3).tobytes()):
        ref = oracle.bwt_encode(v)
        (ok, room, bound), out = run(exe, tmp_path, "benc", v, 0)
        assert ok == 1 and room == len(ref) == len(v) + 2 * (len(v) // 32768) and bound == len(v)
        assert out == ref.tobytes()
        (ok, room, _), back = run(exe, tmp_path, "bdec", ref, 0)
        assert ok == 1 and back == bytes(v)
    bad = oracle.bwt_encode(files["alice29.txt"]).copy()
    bad[32769] |= 0x80  # row index past the block
    (ok, _, _), _ = run(exe, tmp_path, "bdec", bad, 0)
    assert ok == 0


def _goldens():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "encode_traces.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


def test_resumable_encoder_hands_on_what_the_reference_writes(oracle):
    """rcx_estream_*: AdaptiveRangeEncoder<T>::encode fed piece by piece (cpprcoder.h:697-720).  After every call the bytes
    handed on so far must be the prefix of the stream the reference's sink holds by then (tests/golden/encode_traces.json,
    from the reference build; the oracle for the bytes), the results its results -- also when its sink fills in a symbol
    (cpprcoder.h:708-711) or only in finish() (cpprcoder.h:716)."""
    import hashlib

    import trace_cases
    from cpprcoder_amd import rcx
    want = _goldens()
    ctx = rcx.Context(0)
    try:
        for name, data, piece, cap in trace_cases.cases():
            w = want[name]
            n = len(data)
            (ost, orq), osink, osizes = oracle.adaptive_encode_trace(data, piece, cap)
            assert osizes == w["sink_sizes"], name
            full_cap = None if cap is None else (cap + 15) // 16 * 16  # a MemoryStream rounds its capacity up (cpprcoder.h:975)
            es = ctx.estream(n)
            sink = bytearray(n.to_bytes(4, "little"))  # initialize() writes the header through write()
            sizes = [len(sink)]
            st, rq = rcx.OK, 0
            pieces = [b""] if n == 0 else [bytes(data[at: at + piece]) for at in range(0, n, piece)]
            for k, pc in enumerate(pieces):
                room = None if full_cap is None else max(full_cap - len(sink), 0)
                st, rq, body, tail = es.encode(pc, room)
                sink += body
                if tail:  # the final low goes through write(), which grows the sink (cpprcoder.h:756-761, :1031-1045)
                    sink += tail
                sizes.append(len(sink))
                if st == rcx.PENDING and rq != n - min(n, (k + 1) * piece):
                    break  # the sink filled
            assert [st, rq] == w["status"], (name, st, rq)
            assert sizes == w["sink_sizes"], name
            shown = bytes(sink) if cap is None else bytes(sink)[:cap]  # (the generator kept min(size, requested capacity) bytes)
            assert hashlib.sha256(shown).hexdigest() == w["sink_sha256"], name
            assert shown == osink, name
            es.close()
        # rewind: a call taken back leaves no trace
        data = workloads.zipf(9000, 3).tobytes()
        es = ctx.estream(len(data))
        first = es.encode(data[:3000])
        second = es.encode(data[3000:6000])
        es.rewind()
        again = es.encode(data[3000:6000])
        assert again == second
        third = es.encode(data[6000:])
        whole = len(data).to_bytes(4, "little") + first[2] + second[2] + third[2] + third[3]
        assert third[0] == rcx.OK and whole == oracle.adaptive_encode(data)[1]
        # more than was declared: refused (the reference asserts, cpprcoder.h:700)
        es2 = ctx.estream(10)
        st = rcx.lib().rcx_estream_encode(es2._h, data, 11, None, 0, 0, __import__("ctypes").byref(__import__("ctypes").c_uint64()), None, None)
        assert st == rcx.E_ARG
    finally:
        ctx.close()


def test_facade_encoder_writes_to_its_sink_as_it_goes(exe, tmp_path):
    """The same through the C++ facade: the sink's size after every encode() call is the reference's."""
    import trace_cases
    want = _goldens()
    for name, data, piece, cap in trace_cases.cases():
        if piece == 1 and len(data) > 100:
            continue  # (a launch per byte: covered by the C ABI test above)
        w = want[name]
        n = len(data)
        sink_cap = cap if cap is not None else n + n // 32 + 1024
        if cap is None and w["sink_sizes"][-1] > sink_cap:
            sink_cap = w["sink_sizes"][-1] + 16
        p_src, p_dst = tmp_path / "in.bin", tmp_path / "out.bin"
        p_src.write_bytes(bytes(data))
        p = subprocess.run([exe, "enct", str(p_src), str(p_dst), str(sink_cap), str(piece)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (name, p.returncode, p.stderr)
        lines = p.stdout.strip().split("\n")
        st, rq, size, _ = [int(x) for x in lines[0].split()]
        sizes = [int(x) for x in lines[1].split()]
        if cap is None:
            # (the generator's sink had the oracle's default capacity; the bytes and sizes do not depend on it as long as it is large enough)
            assert [st, rq] == w["status"] and sizes == w["sink_sizes"], name
        else:
            assert [st, rq] == w["status"] and sizes == w["sink_sizes"], name
        import hashlib
        got = p_dst.read_bytes()
        assert hashlib.sha256(got if cap is None else got[:cap]).hexdigest() == w["sink_sha256"], name
