"""Where a block of rcx_bwt_fwd_k spends its time (diagnostic build: -DRCX_BWT_STAMP).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DRCX_BWT_STAMP -o build/librcx_stampbwt.so \\
          cpprcoder_amd/csrc/rcx_api.hip cpprcoder_amd/csrc/rcx_comm.hip -L/opt/rocm/lib -lrccl
    RCX_LIBRARY=build/librcx_stampbwt.so python tools/diag/stamp_bwt.py [workload ...]

s_memtime counts shader clocks here (2.4 GHz under load).  Thread 0 of every workgroup sums the time between phase marks
over its blocks; the numbers below are per block (sum / blocks), in microseconds at 2.4 GHz.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch  # noqa: E402
from cpprcoder_amd import rcx, workloads  # noqa: E402

NAMES = ("block in", "start: two passes", "regrouping (all)", "full passes (all rounds)", "collect", "list passes, long list",
         "list passes, short list", "place", "loop control", "column gathered", "block out", "between blocks")
n = 1 << 28
ctx = rcx.Context(0)
L = rcx.lib()
for wl in (sys.argv[1:] or ["uniform", "canterbury"]):
    src = torch.from_numpy(workloads.by_name(wl, n, 12345)).cuda()
    enc = torch.empty(rcx.bwt_encode_bound(n), dtype=torch.uint8, device="cuda")
    ctx.bwt_encode_device(src, enc)
    ctx.sync_status()
    o = (C.c_ulonglong * 16)()
    L.rcx_debug_bwt_stamps(o, 1)
    ctx.bwt_encode_device(src, enc)
    ctx.sync_status()
    L.rcx_debug_bwt_stamps(o, 1)
    blocks = n // 32768
    total = sum(o[i] for i in range(12))
    print(f"{wl}: {blocks} blocks, {total / blocks / 2400:.1f} us per block in all")
    for i, name in enumerate(NAMES):
        print(f"  {name:28s} {o[i] / blocks / 2400:7.2f} us")
