#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference.

Runs only in the build container: it needs oracle/_ref/libcpprcoder_ref.so, i.e. the
unmodified /root/reference/cpprcoder.h compiled by ``make -C oracle ref``.  The
outputs are data (inputs, or how to regenerate them, and the reference's output
bytes / sizes / hashes); no reference source goes into them.  The GPU box and CI
only ever read the JSON.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from cpprcoder_amd import workloads  # noqa: E402


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def main() -> None:
    oracle_lib.build_oracle()
    ref = oracle_lib.reference()
    if ref is None:
        raise SystemExit("oracle/_ref/libcpprcoder_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    fnv = oracle_lib.fnv1a64

    # ---- small known-answer vectors, full stream bytes -------------------
    small = [b"", b"\x00", b"\xff", b"ab", b"abcabcabcabc", b"hello world", b"\xff" * 16, b"\x00" * 16,
             bytes(range(256)), bytes(range(255, -1, -1)), b"\x00\xff" * 40, b"The quick brown fox jumps over the lazy dog"]
    kat = []
    for v in small:
        (st, rq), out, size = ref.adaptive_encode(v)
        ok, sout, ssize = ref.static_encode(v)
        kat.append({"input_hex": v.hex(), "adaptive_hex": out.hex(), "adaptive_status": [st, rq],
                    "static_ok": ok, "static_size": ssize, "static_sha256": sha(sout), "static_tail_hex": sout[516:].hex()})

    # ---- generated inputs, sizes + hashes ---------------------------------
    gen = {
        "A*65535": b"A" * 65535, "A*65536": b"A" * 65536, "A*65537": b"A" * 65537,
        "ff*70000": b"\xff" * 70000,
        "uniform(65536,12345)": workloads.uniform(65536, 12345).tobytes(),
        "uniform(100000,7)": workloads.uniform(100000, 7).tobytes(),
        "zipf(65536,12345)": workloads.zipf(65536, 12345).tobytes(),
        "runs(200000,7)": workloads.runs(200000, 7).tobytes(),
    }
    big = {}
    for name, v in gen.items():
        (st, rq), out, size = ref.adaptive_encode(v)
        ok, sout, ssize = ref.static_encode(v)
        big[name] = {"n": len(v), "input_sha256": sha(v), "adaptive_size": size, "adaptive_sha256": sha(out),
                     "adaptive_fnv1a64": "%016x" % fnv(out), "adaptive_status": [st, rq],
                     "static_ok": ok, "static_size": ssize, "static_sha256": sha(sout)}

    # ---- Canterbury whole files (README.md:16-46 ratios) ------------------
    files = workloads.canterbury_files()
    cant = {}
    for name in workloads.CANTERBURY_ORDER:
        v = files[name]
        (st, rq), out, size = ref.adaptive_encode(v)
        ok, sout, ssize = ref.static_encode(v)
        cant[name] = {"n": len(v), "adaptive_size": size, "adaptive_sha256": sha(out), "adaptive_fnv1a64": "%016x" % fnv(out),
                      "adaptive_ratio": round(size / len(v), 6),
                      "static_size": ssize, "static_sha256": sha(sout), "static_fnv1a64": "%016x" % fnv(sout),
                      "static_ratio": round(ssize / len(v), 6)}

    # ---- blocked runs: per-block size + fnv --------------------------------
    blocks = []
    plan = [(4096, 256), (16384, 128), (65536, 64), (262144, 16)]
    for wl in ("uniform", "zipf", "canterbury", "runs"):
        for block, nblocks in plan:
            n = block * nblocks - (block // 3 if wl != "uniform" else 0)  # ragged last block except for uniform
            data = workloads.by_name(wl, n, 12345)
            for coder, cname in ((0, "adaptive"), (1, "static")):
                if coder == 1 and block != 65536:
                    continue
                slots, sizes = ref.encode_blocks(data, block, coder=coder, threads=8)
                back, ok = ref.decode_blocks(slots, sizes, block, n, coder=coder, threads=8)
                assert ok and np.array_equal(back, data), (wl, block, cname)
                blocks.append({"workload": wl, "seed": 12345, "n": n, "block": block, "coder": cname,
                               "input_sha256": sha(data), "sizes": [int(s) for s in sizes],
                               "fnv1a64": ["%016x" % fnv(slots[b, : int(sizes[b])]) for b in range(len(sizes))],
                               "total": int(sizes.astype(np.uint64).sum())})

    # ---- behaviour pins ----------------------------------------------------
    u64k = workloads.uniform(65536, 12345)
    pins = {}
    (st, rq), out, size = ref.adaptive_encode(u64k, sink_capacity=65536)
    pins["overflow_uniform64k_into_65536"] = {"status": [st, rq], "size": size, "prefix_sha256": sha(out)}
    (st, rq), out, size = ref.adaptive_decode(bytes.fromhex("000000000000000000"), 16)
    pins["decode_of_empty_stream"] = {"status": [st, rq], "out_hex": out.hex()}
    (st, rq), out, size = ref.adaptive_decode(b"\x01\x00\x00", 16)
    pins["decode_short_input"] = {"status": [st, rq], "out_hex": out.hex()}
    junk = b"\x40\x00\x00\x00" + workloads.uniform(200, 99).tobytes()
    (st, rq), out, size = ref.adaptive_decode(junk, 64)
    pins["decode_junk_64"] = {"input_hex": junk.hex(), "status": [st, rq], "out_hex": out.hex()}
    junk2 = b"\x40\x00\x00\x00\x00" + b"\xff" * 100
    (st, rq), out, size = ref.adaptive_decode(junk2, 64)
    pins["decode_allff_64"] = {"input_hex": junk2.hex(), "status": [st, rq], "out_hex": out.hex()}
    (st, rq), comp, size = ref.adaptive_encode(u64k[:3000])
    (st, rq), out, size = ref.adaptive_decode(comp[:1500], 3000)
    pins["decode_truncated"] = {"status": [st, rq], "size": size, "out_sha256": sha(out)}
    (st, rq), out, size = ref.adaptive_decode(comp, 1000)
    pins["decode_sink_full"] = {"status": [st, rq], "size": size, "out_sha256": sha(out)}

    # sink scripts
    scripts = [
        [(0, 0), (3, 5), (2, 4), (3, 6), (2, 2000), (3, 1), (5, 10), (4, 8), (4, 5000), (5, 0), (3, 9)],
        [(1, 0), (3, 1)] + [(3, i) for i in range(16)] + [(2, 4), (2, 20000), (2, 1)],
        [(1, 65536), (2, 4), (5, 65536), (3, 1), (2, 4), (2, 4)],
        [(1, -5), (2, 16), (2, 1), (2, 16384), (2, 1), (4, 100), (4, 40000), (5, 50000), (2, 0)],
    ]
    pins["stream_scripts"] = [{"ops": s, "trace": ref.stream_script(s)} for s in scripts]

    # model probes (incl. the halving at total = 2**24)
    probes = []
    for label, syms in (("none", np.zeros(0, np.uint8)), ("zipf50k", workloads.zipf(50000, 3)),
                        ("uniform_to_halving", workloads.uniform((1 << 24) - 256 + 5000, 11)),
                        ("skew_to_halving", np.minimum(workloads.zipf((1 << 24) - 256, 5), 3).astype(np.uint8))):
        total, freq, cum, _, _ = ref.model_probe(syms, [])
        targets = sorted(set([0, 1, 15, 16, 17, 255, 256, total // 2, max(total - 1, 0), total, total + 1, 0xFFFFFFFF]
                             + [int(x) for x in np.random.RandomState(1).randint(0, total, 32)]))
        total, freq, cum, fcount, fcode = ref.model_probe(syms, targets)
        probes.append({"label": label, "n": len(syms), "input_sha256": sha(syms), "total": total,
                       "freq_sha256": sha(freq.tobytes()), "cum_sha256": sha(cum.tobytes()),
                       "targets": targets, "found_count": [int(x) for x in fcount], "found_code": [int(x) for x in fcode]})
    pins["model_probes"] = probes

    out = {"generator": "tests/golden/make_golden.py", "source": "oracle/_ref/libcpprcoder_ref.so (unmodified /root/reference/cpprcoder.h)",
           "kat": kat, "generated": big, "canterbury": cant, "pins": pins}
    with open(os.path.join(HERE, "adaptive_static_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(HERE, "blocks.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "blocks": blocks}, f, separators=(",", ":"))
    print("wrote", len(kat), "kats,", len(big), "generated,", len(cant), "canterbury,", len(blocks), "block tables")


if __name__ == "__main__":
    main()
