import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bwt_cases
from cpprcoder_amd import rcx
ctx = rcx.Context(0)
for p in (2, 4, 64, 1024, 16384):
    blocks = np.concatenate([bwt_cases.periodic(p, 1000 + i, 256) for i in range(64)])
    data = torch.from_numpy(np.tile(blocks, 16)).cuda()   # 1024 blocks = 32 MiB
    enc = torch.empty(rcx.bwt_encode_bound(data.numel()), dtype=torch.uint8, device='cuda')
    ctx.bwt_encode_device(data, enc); ctx.sync_status()
    torch.cuda.synchronize(); t=time.time()
    ctx.bwt_encode_device(data, enc); ctx.sync_status(); torch.cuda.synchronize()
    dt=time.time()-t
    print(f"period {p}: 1024 periodic blocks in {dt*1e3:.1f} ms = {dt/1024*1e6:.0f} us per block amortised; ties {ctx.bwt_last_ties()}", flush=True)
