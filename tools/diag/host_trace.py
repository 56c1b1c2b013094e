"""One traced host-buffer encode + decode of the uniform GiB (RCX_HOST_TRACE=1): the timeline of rcx_host.hpp's stages on stderr."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cpprcoder_amd import rcx, workloads

n, block = 1 << 30, 65536
data = workloads.by_name("uniform", n, 12345)
ctx = rcx.Context(0)
dst = np.zeros(rcx.encode_bound(n, block), dtype=np.uint8)
out = np.zeros(n, dtype=np.uint8)
offsets = np.zeros(rcx.block_count(n, block) + 1, dtype=np.uint64)
for rep in range(2):
    size = ctx.encode_blocks_into(data, block, dst, offsets)
    ctx.decode_blocks_into(dst, size, offsets, block, out)
os.environ["RCX_HOST_TRACE"] = "1"
print("=== encode", file=sys.stderr, flush=True)
size = ctx.encode_blocks_into(data, block, dst, offsets)
print("=== decode", file=sys.stderr, flush=True)
ctx.decode_blocks_into(dst, size, offsets, block, out)
assert np.array_equal(out, data)
ctx.close()
