#!/usr/bin/env python3
"""Block-size / workload sweep on one GPU (BASELINE.json configs[2], [4]-shaped data on one GPU, configs[5]).

    python tools/sweep.py [--bytes N] [--out gpurun_out/sweep.jsonl]

For every workload and block size: encode + decode of the same resident buffer, median of 3,
round trip verified, ratio = compressed / original.  One JSON line per point.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=1 << 30)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.jsonl"))
    ap.add_argument("--workloads", default="uniform,canterbury,zipf")
    ap.add_argument("--blocks", default="4096,8192,16384,32768,65536,131072,262144")
    ap.add_argument("--coder", type=int, default=0, help="0 adaptive, 1 static, 2 rANS (one state), 3 rANS (eight states)")
    args = ap.parse_args()
    import torch
    from cpprcoder_amd import rcx, workloads
    n = args.bytes
    ctx = rcx.Context(0)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for wl in args.workloads.split(","):
            t0 = time.time()
            src = torch.from_numpy(workloads.by_name(wl, n, 12345)).cuda()
            gen_s = time.time() - t0
            for block in [int(b) for b in args.blocks.split(",")]:
                nblocks = rcx.block_count(n, block)
                dst = torch.empty(rcx.encode_bound(n, block, args.coder), dtype=torch.uint8, device="cuda")
                offs = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
                out = torch.empty(n, dtype=torch.uint8, device="cuda")
                ctx.reserve(n, block, args.coder)
                enc_ms, dec_ms = [], []
                for rep in range(4):
                    ctx.set_timing(True)
                    ctx.get_timing(reset=True)
                    ctx.encode_blocks_device(src, block, dst, offs, coder=args.coder)
                    ctx.decode_blocks_device(dst, dst.numel(), offs, n, block, out, coder=args.coder)
                    ctx.sync_status()
                    t = ctx.get_timing(reset=True)
                    if rep:
                        enc_ms.append(t["encode"]["ms"] + t["scan"]["ms"] + t["scatter"]["ms"])
                        dec_ms.append(t["decode"]["ms"])
                ok = bool(torch.equal(out, src))
                # launch shape of the adaptive coder (rcx_api.hip: encode_lanes / decode_quads; SURVEY.md section 8(d) config 5
                # asks for waves per CU and LDS bytes per wave next to the rates)
                shape = {}
                if args.coder == 0:
                    cus = 256
                    pow2 = lambda x: 1 << max(0, (int(x) - 1).bit_length())
                    lanes = min(64, pow2(-(-nblocks // cus)))
                    quads = min(16, pow2(-(-nblocks // (4 * cus))))
                    enc_wgs, dec_waves = -(-nblocks // lanes), -(-nblocks // quads)
                    shape = {"enc_blocks_per_workgroup": lanes, "enc_workgroups": enc_wgs, "enc_waves_per_cu": round(5 * min(enc_wgs, cus) / cus, 2),
                             "enc_lds_bytes_per_wave": 146432 // 5, "dec_blocks_per_wave": quads, "dec_waves": dec_waves,
                             "dec_waves_per_cu": round(min(dec_waves, 8 * cus) / cus, 2), "dec_lds_bytes_per_wave": 79872 // 4}
                total = int(offs[-1])
                e, d = sorted(enc_ms)[1], sorted(dec_ms)[1]
                line = {"coder": ("adaptive", "static", "rans", "rans8")[args.coder], "workload": wl, "bytes": n, "block": block, "blocks": nblocks, "ratio": round(total / n, 6),
                        "encode_ms": round(e, 3), "decode_ms": round(d, 3), "encode_MBps": round(n / 1e6 / (e * 1e-3), 1),
                        "decode_MBps": round(n / 1e6 / (d * 1e-3), 1), "roundtrip_MBps": round(n / 1e6 / ((e + d) * 1e-3), 1),
                        "roundtrip_ok": ok, "gen_s": round(gen_s, 1), **shape}
                print(json.dumps(line), flush=True)
                f.write(json.dumps(line) + "\n")
                f.flush()
                del dst, offs, out
    ctx.close()


if __name__ == "__main__":
    main()
