#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (rcx_encode_blocks / rcx_decode_blocks): pageable host memory in,
pageable host memory out, copies and synchronisation included.  DESIGN.md section 7; never bench.py's `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from cpprcoder_amd import rcx, workloads
    n, block = 1 << 30, 65536
    ctx = rcx.Context(0)
    for wl in ("uniform", "zipf"):
        data = workloads.by_name(wl, n, 12345)
        ctx.encode_blocks(data[: 1 << 24], block)  # warm up (allocations)
        best_e = best_d = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            payload, offsets = ctx.encode_blocks(data, block)
            t1 = time.perf_counter()
            back = ctx.decode_blocks(payload, offsets, block, capacity=n)
            t2 = time.perf_counter()
            assert np.array_equal(back, data)
            best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
        print(json.dumps({"workload": wl, "bytes": n, "block": block, "ratio": round(len(payload) / n, 6),
                          "host_encode_MBps": round(n / 1e6 / best_e, 1), "host_decode_MBps": round(n / 1e6 / best_d, 1),
                          "note": "pageable numpy buffers, hipMemcpy in and out, python wrapper allocations included"}), flush=True)
    # the block sort through its host-buffer calls (rcx_bwt_encode / rcx_bwt_decode)
    for wl in ("uniform", "canterbury"):
        data = workloads.by_name(wl, n, 12345)
        ctx.bwt_encode(data[: 1 << 24])
        best_e = best_d = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            enc = ctx.bwt_encode(data)
            t1 = time.perf_counter()
            back = ctx.bwt_decode(enc)
            t2 = time.perf_counter()
            assert np.array_equal(back, data)
            best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
        print(json.dumps({"stage": "blksort", "workload": wl, "bytes": n, "host_forward_MBps": round(n / 1e6 / best_e, 1),
                          "host_inverse_MBps": round(n / 1e6 / best_d, 1),
                          "note": "pageable numpy buffers, hipMemcpy in and out, python wrapper allocations and its canary check included"}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
