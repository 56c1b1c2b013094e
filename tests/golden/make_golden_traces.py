#!/usr/bin/env python3
"""Generate tests/golden/encode_traces.json from the REAL reference (cpprcoder.h:678-802): AdaptiveRangeEncoder<T>::encode
called in pieces, and how many bytes the sink (a MemoryStream) holds after initialize() and after every call -- what a
caller that looks at its sink between calls sees.  Runs only in the build container (needs oracle/_ref/libcpprcoder_ref.so).

    python tests/golden/make_golden_traces.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
import trace_cases  # noqa: E402


def main() -> None:
    oracle_lib.build_oracle()
    ref = oracle_lib.reference()
    if ref is None or ref._aenct is None:
        raise SystemExit("oracle/_ref/libcpprcoder_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    out = {"cases": []}
    for name, data, piece, cap in trace_cases.cases():
        (st, rq), sink, sizes = ref.adaptive_encode_trace(data, piece, cap)
        out["cases"].append({"name": name, "n": len(data), "input_sha256": hashlib.sha256(bytes(data)).hexdigest(), "piece": piece,
                             "sink_capacity": cap, "status": [st, rq], "sink_sizes": sizes, "sink_sha256": hashlib.sha256(sink).hexdigest()})
        print(name, piece, cap, (st, rq), sizes[:6], "...", sizes[-1], flush=True)
    with open(os.path.join(HERE, "encode_traces.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
