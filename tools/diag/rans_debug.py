#!/usr/bin/env python3
"""Diagnostic: rANS stream calls on a few small inputs, printing what comes back (tools/diag, not a test)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib  # noqa: E402
from cpprcoder_amd import rcx, workloads  # noqa: E402

ctx = rcx.Context(0)
o = oracle_lib.oracle()
for v in (b"\x00", b"ab", b"hello world", bytes(range(256)), workloads.zipf(5000, 1).tobytes(), workloads.zipf(70000, 2).tobytes()):
    for coder in (2, 3):
        ref = o.rans_encode(v, coder == 3)
        st, rq, comp = ctx.stream_encode(v, coder=coder)
        st2, rq2, back = ctx.stream_decode(ref, max(len(v), 16), coder=coder)
        print(len(v), coder, "enc", st, comp == ref, len(comp), len(ref), "dec", st2, rq2, back == v, flush=True)
        if back != v and st2 == 0:
            a = np.frombuffer(back, np.uint8)
            b = np.frombuffer(v, np.uint8)
            bad = np.flatnonzero(a[: len(b)] != b)
            print("   first mismatches at", bad[:10], flush=True)
