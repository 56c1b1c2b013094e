import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from cpprcoder_amd import rcx
n, block = 1 << 30, 65536
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
ctx = rcx.Context(0)
dst = torch.empty(rcx.encode_bound(n, block), dtype=torch.uint8, device="cuda")
offs = torch.zeros(n // block + 1, dtype=torch.int64, device="cuda")
for _ in range(2):
    ctx.encode_blocks_device(src, block, dst, offs)
ctx.sync_status()
out = (C.c_ulonglong * 16)()
print("rc", rcx.lib().rcx_debug_stamps(out))
for w in range(6):
    print("wave", w, "total", out[2 * w], "barrier-wait", out[2 * w + 1], "per step total %.1f wait %.1f" % (out[2 * w] / 65536, out[2 * w + 1] / 65536))
