"""The CPU restatement of the reference's rANS coders (oracle/rans_oracle.c; cppans.h:497-649) against the golden
vectors made from the real reference (tests/golden/rans.json), and against the reference build itself when
oracle/_ref/libcppans_ref.so is present.  The reference publishes no rANS numbers: its compiled output is the pin."""
import hashlib

import numpy as np
import pytest

import oracle_lib
from cpprcoder_amd import workloads


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def generated():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_golden_rans", os.path.join(os.path.dirname(__file__), "golden", "make_golden_rans.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.GENERATED


def test_rans_kats(oracle, golden):
    for k in golden["rans"]["kat"]:
        v = bytes.fromhex(k["input_hex"])
        for simd, key in ((False, "rans_hex"), (True, "rans8_hex")):
            comp = oracle.rans_encode(v, simd)
            assert comp.hex() == k[key], (v, simd)
            ok, back = oracle.rans_decode(comp, len(v), simd)
            assert ok and back == v
            # stream layout (cppans.h:521-527, :598-604): u32 n, 257 u32 scaled cumulative counts
            assert int.from_bytes(comp[:4], "little") == len(v)
            cum = np.frombuffer(comp[4:1032], np.uint32)
            assert cum[0] == 0 and cum[256] == (4096 if simd else 16384) and np.all(np.diff(cum.astype(np.int64)) >= 0)


def test_rans_generated_and_canterbury(oracle, golden):
    gens = generated()
    for name, g in golden["rans"]["generated"].items():
        v = gens[name]()
        assert sha(v) == g["input_sha256"]
        for simd, key in ((False, "rans"), (True, "rans8")):
            comp = oracle.rans_encode(v, simd)
            assert (len(comp), sha(comp)) == (g[key + "_size"], g[key + "_sha256"]), (name, key)
            ok, back = oracle.rans_decode(comp, len(v), simd)
            assert ok and back == v.tobytes()
    # one repeated byte: encode_simd spends a 16-bit word per symbol (cppans.h:357: (2^16 >> 12 << 16) * 4096 wraps to 0)
    assert golden["rans"]["generated"]["A*65536"]["rans8_size"] == 2 * 65536 + 1032 + 32
    files = workloads.canterbury_files()
    for name, g in golden["rans"]["canterbury"].items():
        v = np.frombuffer(files[name], np.uint8)
        for simd, key in ((False, "rans"), (True, "rans8")):
            comp = oracle.rans_encode(v, simd)
            assert (len(comp), sha(comp)) == (g[key + "_size"], g[key + "_sha256"]), (name, key)


def test_rans_block_tables(oracle, golden):
    for t in golden["rans"]["blocks"]:
        coder = oracle_lib.CODER_RANS if t["coder"] == "rans" else oracle_lib.CODER_RANS8
        data = workloads.by_name(t["workload"], t["n"], t["seed"])
        assert sha(data) == t["input_sha256"]
        slots, sizes = oracle.encode_blocks(data, t["block"], coder=coder, threads=4)
        assert [int(x) for x in sizes] == t["sizes"], (t["workload"], t["block"], t["coder"])
        assert ["%016x" % oracle_lib.fnv1a64(slots[b, : int(sizes[b])]) for b in range(len(sizes))] == t["fnv1a64"]
        back, ok = oracle.decode_blocks(slots, sizes, t["block"], t["n"], coder=coder, threads=4)
        assert ok and np.array_equal(back, data)


def test_rans_damaged_streams_are_refused(oracle):
    v = workloads.zipf(5000, 1)
    for simd in (False, True):
        comp = bytearray(oracle.rans_encode(v, simd))
        ok, _ = oracle.rans_decode(bytes(comp[: len(comp) // 2]), len(v), simd)  # truncated payload
        assert not ok
        bad = bytearray(comp)
        bad[4 + 4 * 256: 4 + 4 * 257] = (12345).to_bytes(4, "little")  # cum[256] is not the probability scale
        ok, _ = oracle.rans_decode(bytes(bad), len(v), simd)
        assert not ok


def test_rans_against_reference_build(oracle, reference):
    if reference is None or reference.ans is None:
        pytest.skip("oracle/_ref/libcppans_ref.so not built here (needs /root/reference)")
    rs = np.random.RandomState(77)
    cases = [workloads.uniform(5000, 1), workloads.zipf(70000, 2), workloads.runs(150000, 3), workloads.canterbury_tiled(100000)[7:],
             np.zeros(1, np.uint8), np.full(30000, 255, np.uint8)]
    for _ in range(120):
        n = int(rs.randint(1, 3000))
        cases.append(rs.randint(0, int(rs.randint(1, 257)), size=n).astype(np.uint8))
    for n in range(1, 24):  # fewer symbols than states
        cases.append(rs.randint(0, 256, size=n).astype(np.uint8))
    for v in cases:
        for simd in (False, True):
            a, b = oracle.rans_encode(v, simd), reference.rans_encode(v, simd)
            assert a == b, (len(v), simd)
            assert oracle.rans_decode(a, len(v), simd) == reference.rans_decode(a, len(v), simd) == (True, v.tobytes())
