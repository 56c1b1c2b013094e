#!/usr/bin/env python3
"""Generate tests/golden/bwt.json from the REAL reference's block sort (blksort.h:440-541).

Runs only in the build container (needs oracle/_ref/libblksort_ref.so: the unmodified /root/reference/blksort.h compiled
by ``make -C oracle ref``).  Outputs are data: for every input of tests/bwt_cases.py (regenerated from integer
arithmetic, so only its hash is stored) and for the Canterbury files the reference harness runs (test/main.cpp:1248-1258,
from the corpus archive the reference ships), the size and hash of what BlkSort::encode wrote and the row index it
stored behind each block.  The reference publishes no block-sort numbers, so these fixtures are the pin.

    python tests/golden/make_golden_bwt.py
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

import bwt_cases  # noqa: E402
import oracle_lib  # noqa: E402
from cpprcoder_amd import workloads  # noqa: E402


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def main() -> None:
    oracle_lib.build_oracle()
    ref = oracle_lib.reference()
    if ref is None or ref.bwt is None:
        raise SystemExit("oracle/_ref/libblksort_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    out = {"block": bwt_cases.BLOCK, "encoded": bwt_cases.ENCODED, "cases": {}, "canterbury": {}}
    for name, data in bwt_cases.cases().items():
        if name.startswith("period "):
            p = int(name.split()[1].rstrip(","))
            assert bwt_cases.primitive_period(data) == p, name
        enc = ref.bwt_encode(data, threads=8)
        assert len(enc) == ref.bwt_encode_bound(len(data))
        back = ref.bwt_decode(enc, threads=8)
        assert np.array_equal(back, data), name
        out["cases"][name] = {"n": len(data), "input_sha256": sha(data), "encoded_size": len(enc),
                              "encoded_sha256": sha(enc), "rows": bwt_cases.rows(enc)}
        print(name, len(data), out["cases"][name]["rows"][:4], flush=True)
    files = workloads.canterbury_files()
    for name in workloads.CANTERBURY_ORDER:
        data = np.frombuffer(files[name], np.uint8)
        enc = ref.bwt_encode(data, threads=8)
        back = ref.bwt_decode(enc, threads=8)
        assert np.array_equal(back, data), name
        out["canterbury"][name] = {"n": len(data), "encoded_size": len(enc), "encoded_sha256": sha(enc),
                                   "rows": bwt_cases.rows(enc)}
        print(name, len(data), len(enc), flush=True)
    with open(os.path.join(HERE, "bwt.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
