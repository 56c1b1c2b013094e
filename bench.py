#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: encode 1 GiB (per GPU) of 64 KiB blocks into the compacted stream
(encode kernel + size prefix + scatter), then decode it back (BASELINE.json configs[1]).
With N > 1 every rank codes its own 1 GiB shard (weak scaling) and the compressed segments
are concatenated on every rank (allgatherv over RCCL) while the decode runs.

Prints ONE JSON line on rank 0.  `value` = uncompressed MB/s through the whole round trip
(all ranks' bytes / max-over-ranks time).  `roofline` prices the dominant kernel against the
HBM roof (this is integer/byte work, no MFMA); `cpu_baseline` times the reference's own CPU
coder (oracle/_ref, or the oracle port when that build did not travel) on this box's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
BLOCK = 65536


def make_workload(name: str, n: int, seed: int, device):
    """Synthetic bytes of the named shape, generated on the host (seeded mt19937) and moved to HBM."""
    import torch
    from cpprcoder_amd import workloads
    t0 = time.time()
    if name == "uniform" and n >= (1 << 28):
        # 1 GiB of mt19937 output takes ~15 s per rank; draw 64 MiB and extend it with a keyed
        # xor-shuffle so every 64 KiB block is still incompressible and distinct.
        base = torch.from_numpy(workloads.uniform(1 << 26, seed)).to(device)
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        data = torch.randint(0, 256, (n,), dtype=torch.uint8, device=device, generator=g)
        reps = n // base.numel()
        data.view(reps, -1).bitwise_xor_(base.unsqueeze(0))
    else:
        data = torch.from_numpy(workloads.by_name(name, n, seed)).to(device)
    return data, time.time() - t0


def cpu_baseline(name: str, seed: int, sample_bytes: int):
    """The reference's CPU coder on a bounded sample of the same workload, all host cores."""
    import oracle_lib
    from cpprcoder_amd import workloads
    oracle_lib.build_oracle()
    chk = oracle_lib.reference() or oracle_lib.oracle()
    cores = os.cpu_count() or 1
    data = workloads.by_name(name, sample_bytes, seed)
    t0 = time.time()
    slots, sizes = chk.encode_blocks(data, BLOCK, threads=cores)
    t1 = time.time()
    back, ok = chk.decode_blocks(slots, sizes, BLOCK, len(data), threads=cores)
    t2 = time.time()
    assert ok and np.array_equal(back, data)
    one = data[: min(len(data), 8 << 20)]
    s0 = time.time()
    s_slots, s_sizes = chk.encode_blocks(one, BLOCK, threads=1)
    s1 = time.time()
    chk.decode_blocks(s_slots, s_sizes, BLOCK, len(one), threads=1)
    s2 = time.time()
    mb = len(data) / 1e6
    return {
        "value": round(mb / (t2 - t0), 2), "unit": "MB/s", "cores": cores, "kind": chk.kind,
        "sample": f"first {len(data) >> 20} MiB of the same workload, {BLOCK >> 10} KiB blocks, encode+decode round trip, "
                  f"{cores} threads over a block range split (fresh coder per block, as test/main.cpp:325-344)",
        "encode_MBps": round(mb / (t1 - t0), 2), "decode_MBps": round(mb / (t2 - t1), 2),
        "one_thread_encode_MBps": round(len(one) / 1e6 / (s1 - s0), 2), "one_thread_decode_MBps": round(len(one) / 1e6 / (s2 - s1), 2),
        "ratio": round(float(sizes.astype(np.uint64).sum()) / len(data), 6),
    }


# The contract is ONE line on stdout.  RCCL prints a version banner to stdout when a communicator is created
# (and other libraries may chat as well): everything written to file descriptor 1 during the run goes to stderr,
# and the JSON line is written to the real stdout at the end.
_REAL_STDOUT = 1


def _stdout_to_stderr() -> None:
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def main() -> None:
    _stdout_to_stderr()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="uniform", choices=["uniform", "zipf", "canterbury", "runs"])
    ap.add_argument("--bytes", type=int, default=1 << 30, help="uncompressed bytes per GPU")
    ap.add_argument("--block", type=int, default=BLOCK)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=256)
    ap.add_argument("--exchange", action="store_true",
                    help="run the N>1 exchange step (allgatherv over RCCL) even with one rank: rehearses that code path on a 1-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cpprcoder_amd import build, parallel, rcx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    build.build()  # no-op when the in-tree librcx.so is current
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    exchange = world > 1 or args.exchange
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    n, block = args.bytes, args.block
    nblocks = rcx.block_count(n, block)
    seed = 12345 + rank  # mt19937(12345 + rank) per shard
    src, gen_s = make_workload(args.workload, n, seed, device)
    ctx = rcx.Context(local)
    ctx.reserve(n, block)
    bound = rcx.encode_bound(n, block)
    # Two compressed buffers: with the exchange on, step i+1 encodes into the other one while the allgatherv of
    # step i is still reading this one (the exchange of a step overlaps its own decode AND the next step's encode).
    nbuf = 2 if (world > 1 or args.exchange) else 1
    comps = [torch.empty(bound + 256, dtype=torch.uint8, device=device) for _ in range(nbuf)]  # +256: segments are padded to 256 B
    offss = [torch.zeros(nblocks + 1, dtype=torch.int64, device=device) for _ in range(nbuf)]
    out = torch.empty(n, dtype=torch.uint8, device=device)
    concat = torch.empty((bound + 256) * world, dtype=torch.uint8, device=device) if exchange else None
    staging = torch.empty((bound + 256) * world, dtype=torch.uint8, device=device) if exchange else None
    side = torch.cuda.Stream(device=device) if exchange else None
    main_stream = torch.cuda.current_stream()
    gather_ms = []
    released = [None] * nbuf  # event: the exchange that read buffer b has finished
    state = {"i": 0, "table": None, "last": 0}

    def step():
        b = state["i"] % nbuf
        state["i"] += 1
        state["last"] = b
        comp, offs = comps[b], offss[b]
        if released[b] is not None:
            main_stream.wait_event(released[b])
        ctx.encode_blocks_device(src, block, comp, offs)
        if not exchange:
            # the decoder takes the block table from HBM: no host round trip inside the step
            ctx.decode_blocks_device(comp, bound, offs, n, block, out)
            return
        total = int(offs[-1])  # the exchange needs the segment size on the host
        side.wait_stream(main_stream)  # (exchanges also queue behind each other on the side stream)
        ctx.decode_blocks_device(comp, total, offs, n, block, out)   # own shard, overlaps the exchange
        with torch.cuda.stream(side):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bases, sizes, works = parallel.allgatherv_segments(comp, total, concat, staging=staging)
            table = parallel.allgather_offsets(offs, bases)
            for w in works:
                w.wait()
            e1.record()
            gather_ms.append((e0, e1))
            state["table"] = table
            released[b] = e1

    def fence():
        torch.cuda.synchronize()
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.sync_status()
    fence()
    gather_ms.clear()
    ctx.set_timing(True)
    ctx.get_timing(reset=True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    ctx.set_timing(False)
    timing = ctx.get_timing(reset=True)
    st, bad = ctx.sync_status(raise_on_error=False)
    roundtrip_ok = bool(torch.equal(out, src)) and st == 0
    comp, offs = comps[state["last"]], offss[state["last"]]
    total = int(offs[-1])
    ratio = total / n

    if exchange:
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
        ok = torch.tensor([1 if roundtrip_ok else 0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        roundtrip_ok = bool(int(ok[0]))
        # the concatenated stream must hold this rank's segment at its base
        bases = [int(x) for x in (state["table"][::nblocks][:world]).cpu()]
        roundtrip_ok = roundtrip_ok and bool(torch.equal(concat[bases[rank]: bases[rank] + total], comp[:total]))

    if rank == 0:
        ms_step = wall * 1e3 / args.steps
        enc_ms = timing["encode"]["ms"] / max(1, timing["encode"]["launches"])
        dec_ms = timing["decode"]["ms"] / max(1, timing["decode"]["launches"])
        scan_ms = timing["scan"]["ms"] / max(1, timing["scan"]["launches"])
        scat_ms = timing["scatter"]["ms"] / max(1, timing["scatter"]["launches"])
        auto = "rcx_dec_quad_k"  # rcx_api.hip: decode_lanes()
        dec_name = {"1": "rcx_dec_adaptive_k", "4": "rcx_dec_quad_k", "8": "rcx_dec_oct_k"}.get(os.environ.get("RCX_LANES_PER_BLOCK", ""), auto)
        enc_name = {"0": "rcx_enc_adaptive_k", "1": "rcx_enc_oct_k", "2": "rcx_enc_mc_k"}.get(os.environ.get("RCX_ENC_VARIANT", ""), "rcx_enc_mc5_k")
        dom, dom_ms = (dec_name, dec_ms) if dec_ms >= enc_ms else (enc_name, enc_ms)
        algo_bytes = (1.0 + ratio) * n  # 1 read + r write per input byte (or r read + 1 write)
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "MB/s encode+decode per GPU on 64 KiB blocks; ratio vs CPU ref",
            "value": round(world * n / 1e6 / (ms_step * 1e-3), 1),
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{n >> 20} MiB synthetic {args.workload} bytes per GPU, {block >> 10} KiB blocks, "
                                   f"encode+decode round trip resident in HBM (BASELINE.json configs[1])",
                       "bytes_per_gpu": n, "block": block, "blocks_per_gpu": nblocks,
                       "parallelism": f"blocks sharded over {world} GPU(s), one process per GPU"
                                      + ("; allgatherv of the compressed segments overlapped with the decode and with the next step's encode" if exchange else "")},
            "roundtrip_ok": roundtrip_ok, "ratio": round(ratio, 6),
            "encode_MBps": round(n / 1e6 / ((enc_ms + scan_ms + scat_ms) * 1e-3), 1),
            "decode_MBps": round(n / 1e6 / (dec_ms * 1e-3), 1),
            "kernel_ms": {"encode": round(enc_ms, 4), "scan": round(scan_ms, 4), "scatter": round(scat_ms, 4), "decode": round(dec_ms, 4)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "note": "per-block coding is a serial chain per symbol; the ceiling that actually binds is "
                                 "blocks_in_flight x clock / cycles_per_symbol (DESIGN.md), not HBM"},
            "workload_gen_s": round(gen_s, 2),
        }
        if exchange and gather_ms:
            line["allgatherv_ms"] = round(sum(a.elapsed_time(b) for a, b in gather_ms) / len(gather_ms), 3)
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(args.workload, seed, min(n, args.cpu_sample_mib << 20))
                line["cpu_baseline"] = cb
                line["gpu_over_cpu"] = round(line["value"] / cb["value"], 1)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        os.write(_REAL_STDOUT, (json.dumps(line) + "\n").encode())

    ctx.close()
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
