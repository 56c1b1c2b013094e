import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from cpprcoder_amd import rcx
n, block = 1 << 30, 65536
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
ctx = rcx.Context(0)
dst = torch.empty(rcx.encode_bound(n, block), dtype=torch.uint8, device="cuda")
offs = torch.zeros(n // block + 1, dtype=torch.int64, device="cuda")
out = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx.encode_blocks_device(src, block, dst, offs)
for _ in range(2):
    ctx.decode_blocks_device(dst, dst.numel(), offs, n, block, out)
ctx.sync_status()
print("ok", bool(torch.equal(out, src)))
o = (C.c_ulonglong * 8)()
print("rc", rcx.lib().rcx_debug_dec_stamps(o))
names = ["glue+issue", "pull", "div+L3+L2", "L1", "L0", "tail"]
tot = sum(o[i] for i in range(6))
for i in range(6):
    print("%-12s %8.1f cycles/step" % (names[i], o[i] / 65536))
print("total %.1f" % (tot / 65536))
