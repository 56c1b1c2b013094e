#!/usr/bin/env python3
"""Generate tests/golden/long_streams.json from the REAL reference: streams and blocks past 1 MiB, up to and through
the table halving at total = 2^24 (cpprcoder.h:1138-1176), and the static coder's second rescale (cpprcoder.h:561-570).

Runs only in the build container (needs oracle/_ref/libcpprcoder_ref.so, see make_golden.py).  Outputs are data:
how to regenerate each input, and the size / hashes of the reference's output.

    python tests/golden/make_golden_long.py
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

NO_HALVING = (1 << 24) - 256  # RCX_MAX_BLOCK: the longest stream whose table is never halved


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def inputs():
    """label -> bytes; tests regenerate the inputs from these recipes."""
    return {
        "uniform(2MiB+77,3)": workloads.uniform((2 << 20) + 77, 3),
        "zipf(NO_HALVING,4)": workloads.zipf(NO_HALVING, 4),
        "uniform(NO_HALVING+5000,11)": workloads.uniform(NO_HALVING + 5000, 11),
        "min(zipf(2^24+70000,5),3)": np.minimum(workloads.zipf((1 << 24) + 70000, 5), 3).astype(np.uint8),
    }


def static_inputs():
    return {
        "zipf(2^24+1000,6)": workloads.zipf((1 << 24) + 1000, 6),
        "runs(3MiB,2)": workloads.runs(3 << 20, 2),
    }


def main() -> None:
    oracle_lib.build_oracle()
    ref = oracle_lib.reference()
    if ref is None:
        raise SystemExit("oracle/_ref/libcpprcoder_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    fnv = oracle_lib.fnv1a64
    out = {"generator": "tests/golden/make_golden_long.py", "source": "oracle/_ref/libcpprcoder_ref.so (unmodified /root/reference/cpprcoder.h)",
           "adaptive": {}, "static": {}, "blocks": []}
    for label, v in inputs().items():
        (st, rq), comp, size = ref.adaptive_encode(v)
        (dst, drq), back, dsize = ref.adaptive_decode(comp, len(v))
        assert (st, rq, dst, drq) == (0, 0, 0, 0) and back == v.tobytes(), label
        out["adaptive"][label] = {"n": len(v), "input_sha256": sha(v), "size": size, "sha256": sha(comp), "fnv1a64": "%016x" % fnv(comp)}
        print(label, len(v), size, flush=True)
    for label, v in static_inputs().items():
        ok, comp, size = ref.static_encode(v)
        dok, back, dsize = ref.static_decode(comp, len(v))
        assert ok and dok and back == v.tobytes(), label
        out["static"][label] = {"n": len(v), "input_sha256": sha(v), "size": size, "sha256": sha(comp), "fnv1a64": "%016x" % fnv(comp)}
        print("static", label, len(v), size, flush=True)
    # blocks past 1 MiB: 2 MiB blocks (ragged last) and one 8 MiB block
    for wl, block, n in (("uniform", 2 << 20, (6 << 20) + 12345), ("canterbury", 2 << 20, (8 << 20) - 1), ("zipf", 8 << 20, (8 << 20) + 5)):
        data = workloads.by_name(wl, n, 777)
        for coder, cname in ((0, "adaptive"), (1, "static")):
            slots, sizes = ref.encode_blocks(data, block, coder=coder, threads=4)
            back, ok = ref.decode_blocks(slots, sizes, block, n, coder=coder, threads=4)
            assert ok and np.array_equal(back, data)
            out["blocks"].append({"workload": wl, "seed": 777, "n": n, "block": block, "coder": cname, "input_sha256": sha(data),
                                  "sizes": [int(s) for s in sizes],
                                  "fnv1a64": ["%016x" % fnv(slots[b, : int(sizes[b])]) for b in range(len(sizes))]})
            print(wl, block, cname, [int(s) for s in sizes], flush=True)
    with open(os.path.join(HERE, "long_streams.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
