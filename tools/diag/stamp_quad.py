"""Where one symbol of rcx_dec_quad_k spends its time (diagnostic build: -DRCX_STAMP_DEC, see README).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRCX_STAMP_DEC -o build/abl/librcx_stampdec.so cpprcoder_amd/csrc/rcx_api.hip cpprcoder_amd/csrc/rcx_comm.hip -L/opt/rocm/lib -lrccl
    RCX_LIBRARY=build/abl/librcx_stampdec.so python tools/diag/stamp_quad.py
"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from cpprcoder_amd import rcx
n, block = 1 << 28, 65536
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
g = o[3]
print("groups", g)
print("symbol 8: leaf read issued -> round 2 done   %.1f cycles" % (o[0] / g))
print("          -> leaf read of symbol 9 issued   %.1f cycles" % (o[1] / g))
print("symbol 9: leaf read issued -> round 2 done   %.1f cycles" % (o[2] / g))
