"""How long the replay of the reference's sort takes for periodic blocks (csrc/rcx_bwt_tie.hpp): 1024 blocks of random
words per period (amortised, 512 replays run at once), and ONE block each of the replay's worst shapes -- a period that
is a long run and a single other byte -- timed alone (nothing to share the device with)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bwt_cases
from cpprcoder_amd import rcx

ctx = rcx.Context(0)


def timed(data_np):
    data = torch.from_numpy(data_np).cuda()
    enc = torch.empty(rcx.bwt_encode_bound(data.numel()), dtype=torch.uint8, device="cuda")
    ctx.bwt_encode_device(data, enc)
    ctx.sync_status()
    torch.cuda.synchronize()
    t = time.time()
    ctx.bwt_encode_device(data, enc)
    ctx.sync_status()
    torch.cuda.synchronize()
    return time.time() - t


for p in (2, 4, 64, 1024, 16384):
    blocks = np.concatenate([bwt_cases.periodic(p, 1000 + i, 256) for i in range(64)])
    dt = timed(np.tile(blocks, 16))  # 1024 blocks = 32 MiB
    print(f"period {p}: 1024 periodic blocks in {dt*1e3:.1f} ms = {dt/1024*1e6:.0f} us per block amortised; ties {ctx.bwt_last_ties()}", flush=True)
cases = bwt_cases.cases()
for name in ("period 16384, a run then b", "period 4096, a run then b", "period 8192, b then a run", "period 16384 (seed 19, 256 symbols)",
             "period 8192, skewed bytes"):
    dt = timed(cases[name])
    print(f"{name}: one block alone {dt*1e3:.1f} ms; ties {ctx.bwt_last_ties()}", flush=True)
