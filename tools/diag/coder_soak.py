"""One-off soak of the four coders against the oracle at BASELINE's full size: every block stream of 1 GiB (64 KiB
blocks) compared byte for byte with the CPU restatement (all host cores), per workload and coder, plus the round trip.
Not part of the test suite (a few minutes on the GPU box's 256 cores).

    python tools/diag/coder_soak.py [bytes]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from cpprcoder_amd import rcx, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
block = 65536
o = oracle_lib.oracle()
ctx = rcx.Context(0)
threads = min(os.cpu_count() or 16, 128)
bad = 0
for wl in ("uniform", "zipf", "canterbury"):
    data = workloads.by_name(wl, n, 12345)
    for coder, name in ((0, "adaptive"), (1, "static"), (2, "rans"), (3, "rans8")):
        t0 = time.time()
        payload, offsets = ctx.encode_blocks(data, block, coder=coder)
        back = ctx.decode_blocks(payload, offsets, block, capacity=n, coder=coder)
        t1 = time.time()
        slots, sizes = o.encode_blocks(data, block, coder=coder, threads=threads)
        t2 = time.time()
        same_sizes = np.array_equal(np.diff(offsets.astype(np.int64)), sizes.astype(np.int64))
        same = same_sizes
        if same_sizes:
            want, _ = o.compact(slots, sizes)
            same = np.array_equal(payload, want)
        rt = np.array_equal(back, data)
        bad += (not same) + (not rt)
        print(f"{wl} {name}: {len(sizes)} blocks, ratio {len(payload) / n:.6f}, gpu {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s, streams equal {same}, round trip {rt}", flush=True)
        del slots, sizes, payload, back
print("MISMATCHES" if bad else "all equal")
sys.exit(1 if bad else 0)
