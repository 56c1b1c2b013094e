"""Run by tests/test_gpu_parity.py in a process of its own with RCX_LIBRARY = the diagnostic build that carries the
superseded kernels (cpprcoder_amd/librcx_variants.so: csrc/variants/rcx_variants.hpp -- the 8-lanes-per-block coders, the
four-wave encoder): each must still produce the oracle's bytes.  The product library does not contain them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from cpprcoder_amd import rcx, workloads  # noqa: E402

assert rcx.LIB_PATH.endswith("librcx_variants.so"), rcx.LIB_PATH
oracle_lib.build_oracle()
oracle = oracle_lib.oracle()
data = workloads.canterbury_tiled(65536 * 37 + 4321)
for block in (4096, 65536):
    slots, sizes = oracle.encode_blocks(data, block, threads=8)
    want_payload, want_offsets = oracle.compact(slots, sizes)
    for enc, dec in (("1", "8"), ("2", "8"), ("1", "4"), ("3", "8")):
        os.environ["RCX_ENC_VARIANT"], os.environ["RCX_LANES_PER_BLOCK"] = enc, dec
        ctx = rcx.Context(0)
        payload, offsets = ctx.encode_blocks(data, block)
        assert np.array_equal(payload, want_payload) and np.array_equal(offsets, want_offsets), (block, enc)
        back = ctx.decode_blocks(payload, offsets, block, capacity=len(data))
        assert np.array_equal(back, data), (block, dec)
        ctx.close()
print("variants ok")
