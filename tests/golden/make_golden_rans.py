#!/usr/bin/env python3
"""Generate tests/golden/rans.json from the REAL reference's rANS coders (cppans.h:497-649).

Runs only in the build container (needs oracle/_ref/libcppans_ref.so: the unmodified /root/reference/cppans.h compiled
by ``make -C oracle ref``).  Outputs are data: inputs (or how to regenerate them) and the reference's output bytes /
sizes / hashes.  "rans" = rANS::encode (one state, 14-bit, bytes), "rans8" = rANS::encode_simd (eight interleaved
states, 12-bit, 16-bit words).  The reference publishes no rANS numbers, so these fixtures are the pin.

    python tests/golden/make_golden_rans.py
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


SMALL = [b"\x00", b"\xff", b"ab", b"abcabcabcabc", b"hello world", b"\xff" * 16, b"\x00" * 17, bytes(range(256)),
         b"\x00\xff" * 40, b"The quick brown fox jumps over the lazy dog", bytes(range(7)) * 3]

GENERATED = {
    "A*65536": lambda: np.full(65536, 65, np.uint8),          # one symbol: encode_simd spends a word per symbol (cppans.h:357 wraps)
    "uniform(65536,12345)": lambda: workloads.uniform(65536, 12345),
    "uniform(100003,7)": lambda: workloads.uniform(100003, 7),
    "zipf(65536,12345)": lambda: workloads.zipf(65536, 12345),
    "runs(200000,7)": lambda: workloads.runs(200000, 7),
    "two symbols 1:70000": lambda: np.concatenate([np.zeros(70000, np.uint8), np.ones(1, np.uint8)]),  # the steal loop of normalize()
    "rare tail": lambda: np.concatenate([workloads.zipf(300000, 3), np.arange(256, dtype=np.uint8)]),
}


def main() -> None:
    oracle_lib.build_oracle()
    ref = oracle_lib.reference()
    if ref is None or ref.ans is None:
        raise SystemExit("oracle/_ref/libcppans_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    fnv = oracle_lib.fnv1a64
    kat = []
    for v in SMALL:
        a, b = ref.rans_encode(v, False), ref.rans_encode(v, True)
        for comp, simd in ((a, False), (b, True)):
            ok, back = ref.rans_decode(comp, len(v), simd)
            assert ok and back == v
        kat.append({"input_hex": v.hex(), "rans_hex": a.hex(), "rans8_hex": b.hex()})
    big = {}
    for name, make in GENERATED.items():
        v = make()
        a, b = ref.rans_encode(v, False), ref.rans_encode(v, True)
        big[name] = {"n": len(v), "input_sha256": sha(v), "rans_size": len(a), "rans_sha256": sha(a), "rans8_size": len(b), "rans8_sha256": sha(b)}
    files = workloads.canterbury_files()
    cant = {}
    for name in workloads.CANTERBURY_ORDER:
        v = np.frombuffer(files[name], np.uint8)
        a, b = ref.rans_encode(v, False), ref.rans_encode(v, True)
        cant[name] = {"n": len(v), "rans_size": len(a), "rans_sha256": sha(a), "rans_ratio": round(len(a) / len(v), 6),
                      "rans8_size": len(b), "rans8_sha256": sha(b), "rans8_ratio": round(len(b) / len(v), 6)}
    blocks = []
    for wl in ("uniform", "zipf", "canterbury", "runs"):
        for block, nblocks in ((4096, 128), (65536, 48), (262144, 6)):
            n = block * nblocks - (block // 3 + 5 if wl != "uniform" else 0)  # ragged last block except for uniform
            data = workloads.by_name(wl, n, 4321)
            for coder, cname in ((oracle_lib.CODER_RANS, "rans"), (oracle_lib.CODER_RANS8, "rans8")):
                slots, sizes = ref.encode_blocks(data, block, coder=coder, threads=8)
                back, ok = ref.decode_blocks(slots, sizes, block, n, coder=coder, threads=8)
                assert ok and np.array_equal(back, data), (wl, block, cname)
                blocks.append({"workload": wl, "seed": 4321, "n": n, "block": block, "coder": cname, "input_sha256": sha(data),
                               "sizes": [int(s) for s in sizes],
                               "fnv1a64": ["%016x" % fnv(slots[b, : int(sizes[b])]) for b in range(len(sizes))],
                               "total": int(sizes.astype(np.uint64).sum())})
    out = {"generator": "tests/golden/make_golden_rans.py", "source": "oracle/_ref/libcppans_ref.so (unmodified /root/reference/cppans.h)",
           "kat": kat, "generated": big, "canterbury": cant, "blocks": blocks}
    with open(os.path.join(HERE, "rans.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", len(kat), "kats,", len(big), "generated,", len(cant), "canterbury,", len(blocks), "block tables;",
          os.path.getsize(os.path.join(HERE, "rans.json")), "bytes")


if __name__ == "__main__":
    main()
