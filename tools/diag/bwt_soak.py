"""One-off soak of the block sort against the oracle: many blocks of several workloads and seeds, GPU output compared
byte for byte with oracle/bwt_oracle.c (16 threads), plus the round trip.  Not part of the test suite (about a minute)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from cpprcoder_amd import rcx, workloads  # noqa: E402

o = oracle_lib.oracle()
ctx = rcx.Context(0)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bad = 0
for wl in ("uniform", "zipf", "canterbury", "runs"):
    for seed in (101, 202, 303):
        if wl == "runs" and seed != 101:
            continue  # (long runs cost the oracle, like the reference, about a second a block)
        data = workloads.by_name(wl, (blocks // 8 if wl == "runs" else blocks) * 32768, seed)
        t0 = time.time()
        enc = ctx.bwt_encode(data)
        t1 = time.time()
        want = o.bwt_encode(data, threads=min(os.cpu_count() or 16, 128))
        t2 = time.time()
        same = np.array_equal(enc, want)
        back = np.array_equal(ctx.bwt_decode(enc), data)
        bad += (not same) + (not back)
        print(f"{wl} seed {seed}: {len(data) // 32768} blocks, gpu {t1 - t0:.2f} s (host buffers), oracle {t2 - t1:.1f} s, same {same}, round trip {back}", flush=True)
print("MISMATCHES" if bad else "all equal")
sys.exit(1 if bad else 0)
