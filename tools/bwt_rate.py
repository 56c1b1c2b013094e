#!/usr/bin/env python3
"""Block sort (rcx_bwt_*) rate on one GPU: forward and inverse transform of a resident buffer per workload, kernel time
from HIP events on the stream (median of 3 after a warm-up), round trip verified.  One JSON line per workload.

    python tools/bwt_rate.py [--bytes N] [--workloads uniform,zipf,canterbury,runs] [--out gpurun_out/bwt_rate.jsonl]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=1 << 30)
    ap.add_argument("--workloads", default="uniform,zipf,canterbury,runs")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bwt_rate.jsonl"))
    args = ap.parse_args()
    import torch
    from cpprcoder_amd import rcx, workloads
    n = args.bytes
    ctx = rcx.Context(0)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for wl in args.workloads.split(","):
            src = torch.from_numpy(workloads.by_name(wl, n, 12345)).cuda()
            enc = torch.empty(rcx.bwt_encode_bound(n), dtype=torch.uint8, device="cuda")
            out = torch.empty(n, dtype=torch.uint8, device="cuda")
            fwd, inv = [], []
            for rep in range(4):
                ctx.set_timing(True)
                ctx.get_timing(reset=True)
                ctx.bwt_encode_device(src, enc)
                ctx.bwt_decode_device(enc, enc.numel(), out)
                ctx.sync_status()
                t = ctx.get_timing(reset=True)
                if rep:
                    fwd.append(t["bwt_forward"]["ms"])
                    inv.append(t["bwt_inverse"]["ms"])
            a, b = sorted(fwd)[1], sorted(inv)[1]
            line = {"stage": "blksort", "workload": wl, "bytes": n, "blocks": n // rcx.BWT_BLOCK, "forward_ms": round(a, 3), "inverse_ms": round(b, 3),
                    "forward_MBps": round(n / 1e6 / (a * 1e-3), 1), "inverse_MBps": round(n / 1e6 / (b * 1e-3), 1),
                    "ties": ctx.bwt_last_ties(), "roundtrip_ok": bool(torch.equal(out, src))}
            print(json.dumps(line), flush=True)
            f.write(json.dumps(line) + "\n")
            del src, enc, out
    ctx.close()


if __name__ == "__main__":
    main()
