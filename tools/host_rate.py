#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (rcx_encode_blocks / rcx_decode_blocks / rcx_bwt_encode /
rcx_bwt_decode): pageable host memory in, pageable host memory out, everything between the call and its return
included.  The buffers are the caller's and exist before the clock starts, as in the reference's harness
(test/main.cpp:321-336); `fresh` = the destination's pages have never been touched (the first call of a program),
`reused` = they have (every later call).  DESIGN.md section 7; never bench.py's `value`.

    python tools/host_rate.py [--modes direct,staged,register] [--movers 2] [--workloads uniform,zipf] [--blksort]
Each mode runs in its own process (the library reads RCX_HOST_* once per context).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(args):
    import numpy as np
    from cpprcoder_amd import rcx, workloads
    n, block = args.bytes, args.block
    coder = ("adaptive", "static", "rans", "rans8").index(args.coder)
    ctx = rcx.Context(0)
    tag = {"mode": os.environ.get("RCX_HOST_MODE", "default"), "movers": os.environ.get("RCX_HOST_MOVERS", "default"),
           "piece_MiB": os.environ.get("RCX_HOST_PIECE_MIB", "default"), "serial": bool(os.environ.get("RCX_HOST_SERIAL"))}
    nblocks = rcx.block_count(n, block)
    for wl in args.workloads.split(","):
        data = workloads.by_name(wl, n, 12345)
        if args.pinned:  # the caller's buffers pinned by the caller (here: torch's pinned allocator = hipHostMalloc)
            import torch
            pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            pin.numpy()[:] = data
            data = pin.numpy()
        warm_dst = np.zeros(rcx.encode_bound(1 << 26, block, coder), dtype=np.uint8)
        ctx.encode_blocks_into(data[: 1 << 26], block, warm_dst, np.zeros(rcx.block_count(1 << 26, block) + 1, dtype=np.uint64), coder)  # allocations, streams
        rows = {}
        for state in ("fresh", "reused", "reused", "reused"):
            if state == "fresh" or "dst" not in rows:
                if args.pinned:
                    import torch
                    keep = (torch.empty(rcx.encode_bound(n, block, coder), dtype=torch.uint8, pin_memory=True), torch.empty(n, dtype=torch.uint8, pin_memory=True))
                    dst, out = keep[0].numpy(), keep[1].numpy()
                else:
                    dst = np.empty(rcx.encode_bound(n, block, coder), dtype=np.uint8)
                    out = np.empty(n, dtype=np.uint8)
                offsets = np.zeros(nblocks + 1, dtype=np.uint64)
                rows["dst"] = True
            t0 = time.perf_counter()
            size = ctx.encode_blocks_into(data, block, dst, offsets, coder)
            t1 = time.perf_counter()
            got = ctx.decode_blocks_into(dst, size, offsets, block, out, coder)
            t2 = time.perf_counter()
            assert got == n and np.array_equal(out, data)
            e, d = rows.get(state, (1e9, 1e9))
            rows[state] = (min(e, t1 - t0), min(d, t2 - t1))
        print(json.dumps(dict(tag, pinned_by_caller=bool(args.pinned), workload=wl, coder=args.coder, bytes=n, block=block, ratio=round(size / n, 6),
                              host_encode_MBps=round(n / 1e6 / rows["reused"][0], 1), host_decode_MBps=round(n / 1e6 / rows["reused"][1], 1),
                              host_encode_MBps_fresh_destination=round(n / 1e6 / rows["fresh"][0], 1),
                              host_decode_MBps_fresh_destination=round(n / 1e6 / rows["fresh"][1], 1),
                              note="pageable numpy buffers made before the clock starts; best of 3 with the destination's pages touched, "
                                   "one run with a destination never touched")), flush=True)
        if args.blksort:
            enc = np.empty(rcx.bwt_encode_bound(n), dtype=np.uint8)
            back = np.empty(n, dtype=np.uint8)
            best_e = best_d = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                m = ctx.bwt_into(True, data, enc)
                t1 = time.perf_counter()
                k = ctx.bwt_into(False, enc[:m], back)
                t2 = time.perf_counter()
                assert k == n and np.array_equal(back, data)
                best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
            print(json.dumps(dict(tag, stage="blksort", workload=wl, bytes=n, host_forward_MBps=round(n / 1e6 / best_e, 1),
                                  host_inverse_MBps=round(n / 1e6 / best_d, 1), note="pageable numpy buffers, reused; best of 3")), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="")
    ap.add_argument("--movers", default="")
    ap.add_argument("--pieces", default="")
    ap.add_argument("--workloads", default="uniform,zipf")
    ap.add_argument("--coder", default="adaptive")
    ap.add_argument("--bytes", type=int, default=1 << 30)
    ap.add_argument("--block", type=int, default=65536)
    ap.add_argument("--blksort", action="store_true")
    ap.add_argument("--pinned", action="store_true", help="the caller's buffers are pinned host memory (hipHostMalloc through torch)")
    ap.add_argument("--serial", action="store_true", help="also the unpipelined path (RCX_HOST_SERIAL=1), for comparison")
    args = ap.parse_args()
    if not args.modes and not args.serial:
        return one(args)
    base = [sys.executable, os.path.abspath(__file__), "--workloads", args.workloads, "--coder", args.coder, "--bytes", str(args.bytes),
            "--block", str(args.block)] + (["--blksort"] if args.blksort else []) + (["--pinned"] if args.pinned else [])
    runs = []
    if args.serial:
        runs.append({"RCX_HOST_SERIAL": "1"})
    for mode in [m for m in args.modes.split(",") if m]:
        for movers in (args.movers.split(",") if args.movers else [""]):
            for piece in (args.pieces.split(",") if args.pieces else [""]):
                env = {"RCX_HOST_MODE": mode}
                if movers:
                    env["RCX_HOST_MOVERS"] = movers
                if piece:
                    env["RCX_HOST_PIECE_MIB"] = piece
                runs.append(env)
    for env in runs:
        rc = subprocess.run(base, env=dict(os.environ, **env)).returncode
        if rc != 0:
            print(json.dumps({"failed": env, "rc": rc}), flush=True)


if __name__ == "__main__":
    main()
