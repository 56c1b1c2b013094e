#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Both forms work for N > 1: typed plainly, this process starts the N ranks itself (one fresh process per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous), never touches the GPU, relays rank 0's line and exits
non-zero if any rank does.

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: encode 1 GiB (per GPU) of 64 KiB blocks into the compacted stream
(encode kernel + size prefix + scatter), then decode it back.  The default workload at N = 1 is
BASELINE.json configs[1] (mt19937(12345) uniform bytes); with N > 1 it is configs[3]'s (Zipf bytes,
mt19937(12345 + rank) per shard): every rank codes its own 1 GiB shard (weak scaling) and the compressed
segments and block tables are concatenated on every rank (rcx_allgatherv_segments: RCCL point to point
over xGMI) while the decode runs.  `config.workload` says which BASELINE.json config the line is.

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
    """Synthetic bytes of the named shape, generated on the host from std::mt19937(seed)
    and moved to HBM.  The CPU baseline below draws the first bytes of the very same stream."""
    import torch
    from cpprcoder_amd import workloads
    t0 = time.time()
    host = workloads.by_name(name, n, seed)
    data = torch.from_numpy(host).to(device)
    return data, host, time.time() - t0


def baseline_config(name: str, n: int, block: int, world: int) -> str:
    """Which BASELINE.json config this run is (0-based index into `configs`), in BASELINE.json's own words."""
    try:
        configs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    except Exception:
        return "BASELINE.json not readable"
    gib = n == (1 << 30)
    if name == "uniform" and gib and block == BLOCK and world == 1:
        return "configs[1]: " + configs[1]
    if name == "canterbury" and gib and block == BLOCK and world == 1:
        return "configs[2]: " + configs[2]
    if name == "zipf" and gib and block == BLOCK and world == 8:
        return "configs[3]: " + configs[3]
    if name == "zipf" and gib and block == BLOCK:
        return f"configs[3] at {world} of its 8 GPUs (1 GiB Zipf shard per GPU, weak scaling): " + configs[3]
    if gib and world == 1 and block != BLOCK:
        return "configs[4] (one point of the block-size sweep): " + configs[4]
    return "none of BASELINE.json's configs (a variation for diagnosis)"


def usable_cpus() -> int:
    """The host cores this process may really use: the visible ones, capped by the container's CPU quota (cgroup v2
    cpu.max, v1 cfs quota) -- the GPU box shows 256 hardware threads and grants 16 cores' worth of time."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(data: np.ndarray, block: int, coder: int = 0, blksort: bool = False):
    """The reference's CPU coder on a bounded sample of the same bytes the GPU coded, all host cores.  With blksort the
    reference's block sort runs in front of it and behind it (BlkSort::encode / decode, as test/main.cpp:961-986)."""
    import oracle_lib
    oracle_lib.build_oracle()
    chk = oracle_lib.reference() or oracle_lib.oracle()
    if blksort and chk.bwt is None:
        chk = oracle_lib.oracle()
    cores = usable_cpus()
    t0 = time.time()
    coded = chk.bwt_encode(data, threads=cores) if blksort else data
    slots, sizes = chk.encode_blocks(coded, block, coder=coder, threads=cores)
    t1 = time.time()
    back, ok = chk.decode_blocks(slots, sizes, block, len(coded), coder=coder, threads=cores)
    if blksort:
        back = chk.bwt_decode(back, threads=cores)
    t2 = time.time()
    assert ok and np.array_equal(back, data)
    one = data[: min(len(data), (2 << 20) if blksort else (8 << 20))]
    s0 = time.time()
    one_coded = chk.bwt_encode(one) if blksort else one
    s_slots, s_sizes = chk.encode_blocks(one_coded, block, coder=coder, threads=1)
    s1 = time.time()
    one_back, _ = chk.decode_blocks(s_slots, s_sizes, block, len(one_coded), coder=coder, threads=1)
    if blksort:
        chk.bwt_decode(one_back)
    s2 = time.time()
    mb = len(data) / 1e6
    return {
        "value": round(mb / (t2 - t0), 2), "unit": "MB/s", "cores": cores, "kind": chk.kind,
        "sample": f"the first {len(data) >> 20} MiB of the very bytes the GPU coded, "
                  + ("block sort of 32 KiB blocks (blksort.h) in front of and behind the coder, " if blksort else "")
                  + f"{block >> 10} KiB blocks, encode+decode round trip, "
                  f"{cores} threads over a block range split (fresh coder per block, as test/main.cpp:325-344)",
        "encode_MBps": round(mb / (t1 - t0), 2), "decode_MBps": round(mb / (t2 - t1), 2),
        "one_thread_encode_MBps": round(len(one) / 1e6 / (s1 - s0), 2), "one_thread_decode_MBps": round(len(one) / 1e6 / (s2 - s1), 2),
        "ratio": round(float(sizes.astype(np.uint64).sum()) / len(data), 6),
    }, sizes


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str], limit_s: float = 1500.0) -> int:
    """`python bench.py --gpus N` typed plainly: start N fresh rank processes (this one has not touched the GPU and never
    does), hand rank 0's stdout through, stderr of all ranks as it comes.  Returns the exit code: 0 only if every rank's is."""
    import subprocess
    import threading
    port = _free_port()
    procs, first_line = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    reader = threading.Thread(target=lambda: first_line.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    code = 0
    alive = set(range(n))
    started = time.time()
    while alive:
        if time.time() - started > limit_s and code == 0:  # a rank that hangs must not hang the whole run
            code = 124
            print(f"bench.py: the ranks did not finish within {limit_s:.0f} s; stopping them", file=sys.stderr)
            for o in alive:
                procs[o].terminate()
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                print(f"bench.py: rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()  # (the very processes started above, by handle)
        time.sleep(0.05)
    reader.join(timeout=10)
    out = (first_line[0] if first_line else b"").decode(errors="replace")
    if code == 0:
        sys.stdout.write(out if out.endswith("\n") or not out else out + "\n")
        sys.stdout.flush()
    return code


def issue_ceiling(rec: dict, kernel: str, symbols_per_chain: int):
    """SURVEY.md section 7, hard part 5: the ceiling that binds these kernels is instruction issue along a block's
    serial chain, not HBM.  From the recorded SQ counters of the kernel (profiles/pmc_traffic.json, the same separate
    rocprofv3 --pmc passes `traffic` comes from): vector instructions per symbol and wave, the wave's cycles per symbol
    (SQ_WAVE_CYCLES counts in units of 4 cycles), and the fraction of those cycles in which a vector instruction of the
    wave was issued (each takes its SIMD 4 cycles, so VALU instructions x 4 / cycles = SQ_INSTS_VALU / SQ_WAVE_CYCLES)."""
    c = rec.get(kernel, {}).get("counters_per_launch", {})
    need = ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES")
    if not all(k in c and c[k] for k in need):
        return None
    waves, symbols = c["SQ_WAVES"], float(symbols_per_chain)
    out = {
        "vector_insts_per_symbol_wave": round(c["SQ_INSTS_VALU"] / waves / symbols, 2),
        "cycles_per_symbol": round(4.0 * c["SQ_WAVE_CYCLES"] / waves / symbols, 1),
        "frac_of_issue": round(c["SQ_INSTS_VALU"] / c["SQ_WAVE_CYCLES"], 4),
        "waves": int(waves),
    }
    if c.get("SQ_ACTIVE_INST_ANY"):
        out["frac_issuing_any_instruction"] = round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    if c.get("SQ_LDS_BANK_CONFLICT") and c.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    return out


def end_to_end(ctx, host_src: np.ndarray, block: int, coder: int):
    """the PCIe-inclusive rate, separately -- never `value`.  The host-buffer calls
    (rcx_encode_blocks / rcx_decode_blocks, csrc/rcx_host.hpp) on pageable host memory that exists before the clock starts,
    as the reference's harness holds its buffers (test/main.cpp:321-336): one pass with a destination whose pages have never
    been touched, then the best of three with the same buffers."""
    from cpprcoder_amd import rcx
    n = len(host_src)
    nblocks = rcx.block_count(n, block)
    dst = np.empty(rcx.encode_bound(n, block, coder), dtype=np.uint8)
    out = np.empty(n, dtype=np.uint8)
    offsets = np.zeros(nblocks + 1, dtype=np.uint64)
    rows = []
    for _ in range(4):
        t0 = time.perf_counter()
        size = ctx.encode_blocks_into(host_src, block, dst, offsets, coder)
        t1 = time.perf_counter()
        got = ctx.decode_blocks_into(dst, size, offsets, block, out, coder)
        t2 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1))
    ok = got == n and bool(np.array_equal(out, host_src))
    enc, dec = min(r[0] for r in rows[1:]), min(r[1] for r in rows[1:])
    link_ms = n / 57e9 * 1e3
    return {
        "encode_MBps": round(n / 1e6 / enc, 1), "decode_MBps": round(n / 1e6 / dec, 1),
        "encode_MBps_fresh_destination": round(n / 1e6 / rows[0][0], 1), "decode_MBps_fresh_destination": round(n / 1e6 / rows[0][1], 1),
        "roundtrip_ok": ok, "bytes": n,
        "what": "rcx_encode_blocks / rcx_decode_blocks: pageable host buffers in and out, chunks of whole blocks with copy in, "
                "kernels and copy out overlapped (csrc/rcx_host.hpp); best of 3 with reused buffers, and the first pass whose "
                "destination pages had never been touched",
        "limit": f"the link: {n >> 20} MiB cross it each way at about 57 GB/s ({link_ms:.1f} ms, both directions at once), plus one "
                 "chunk's kernels (a block's serial chain: 5.6 ms encode, 11.1 ms decode at 64 KiB blocks) and one chunk's copy back "
                 "that nothing can overlap; a fresh destination adds its page faults (host side)",
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["uniform", "zipf", "canterbury", "runs"],
                    help="default: uniform on one GPU (BASELINE.json configs[1]), zipf on several (configs[3])")
    ap.add_argument("--bytes", type=int, default=1 << 30, help="uncompressed bytes per GPU")
    ap.add_argument("--block", type=int, default=BLOCK)
    ap.add_argument("--coder", default="adaptive", choices=["adaptive", "static", "rans", "rans8"],
                    help="adaptive (BASELINE.json's metric) | static | rans | rans8: the same protocol on the sibling coders (diagnosis; not the headline)")
    ap.add_argument("--blksort", action="store_true",
                    help="the reference's block sort (blksort.h, 32 KiB blocks) in front of the encoder and behind the decoder, as its "
                         "harness does in front of zlib / zstd (test/main.cpp:961-986): a pipeline for diagnosis, not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=256)
    ap.add_argument("--exchange", action="store_true",
                    help="run the N>1 exchange step (allgatherv over RCCL) even with one rank: rehearses that code path on a 1-GPU box")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the launcher without a GPU: every rank prints the environment it was started with and exits")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-started ranks get before they are stopped")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch):
        # typed plainly: this process only starts the ranks (it has imported nothing that initialises HIP)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    if args.dry_launch:
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
        mine = {k: os.environ.get(k) for k in keys}
        print(json.dumps({"dry_launch": True, "gpus": args.gpus, "env": mine}), file=sys.stdout if mine["RANK"] == "0" else sys.stderr, flush=True)
        if os.environ.get("RCX_BENCH_DRY_FAIL_RANK") == mine["RANK"]:  # (the launcher's test: a rank that fails)
            raise SystemExit(3)
        if os.environ.get("RCX_BENCH_DRY_FAIL_RANK"):
            time.sleep(2.0)  # (the others are still at work when it does)
        return
    _stdout_to_stderr()

    import torch
    import torch.distributed as dist
    from cpprcoder_amd import build, parallel, rcx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # (started by torch.distributed.run with another count: the environment is what counts)
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

    if args.workload is None:
        args.workload = "uniform" if world == 1 else "zipf"
    n, block = args.bytes, args.block
    coder = ("adaptive", "static", "rans", "rans8").index(args.coder)
    m = rcx.bwt_encode_bound(n) if args.blksort else n  # what the coder is handed (blksort.h:426-431)
    nblocks = rcx.block_count(m, block)
    seed = 12345 + rank  # mt19937(12345 + rank) per shard
    src, host_src, gen_s = make_workload(args.workload, n, seed, device)
    ctx = rcx.Context(local)
    ctx.reserve(m, block, coder)
    bound = rcx.encode_bound(m, block, coder)
    sorted_buf = torch.empty(m, dtype=torch.uint8, device=device) if args.blksort else None
    decoded_buf = torch.empty(m, dtype=torch.uint8, device=device) if args.blksort else None
    # Two compressed buffers: with the exchange on, step i+1 encodes into the other one while the exchange of
    # step i is still reading this one (the exchange of a step overlaps its own decode AND the next step's encode).
    nbuf = 2 if exchange else 1
    comps = [torch.empty(bound, dtype=torch.uint8, device=device) for _ in range(nbuf)]
    offss = [torch.zeros(nblocks + 1, dtype=torch.int64, device=device) for _ in range(nbuf)]
    out = torch.empty(n, dtype=torch.uint8, device=device)
    concat = torch.empty(bound * world, dtype=torch.uint8, device=device) if exchange else None
    table = torch.zeros(nblocks * world + 1, dtype=torch.int64, device=device) if exchange else None
    comm = parallel.Comm.from_process_group(local) if exchange else None  # rcx_comm over RCCL (include/rcx.h)
    side = torch.cuda.Stream(device=device) if exchange else None
    main_stream = torch.cuda.current_stream()
    gather_ms = []
    released = [None] * nbuf  # event: the exchange that read buffer b has finished
    state = {"i": 0, "bases": None, "last": 0}

    def step():
        b = state["i"] % nbuf
        state["i"] += 1
        state["last"] = b
        comp, offs = comps[b], offss[b]
        if released[b] is not None:
            main_stream.wait_event(released[b])
        if args.blksort:
            ctx.bwt_encode_device(src, sorted_buf)
        ctx.encode_blocks_device(sorted_buf if args.blksort else src, block, comp, offs, coder=coder)
        if exchange:
            side.wait_stream(main_stream)  # the exchange starts when the encode is done (and queues behind the previous one)
        # the decoder takes the block table from HBM: no host round trip between encode and decode
        ctx.decode_blocks_device(comp, bound, offs, m, block, decoded_buf if args.blksort else out, coder=coder)
        if args.blksort:
            ctx.bwt_decode_device(decoded_buf, m, out)
        if exchange:
            # one call: sizes all-gather (its one host sync waits for the encode only), then every segment and table
            # part point to point straight into place -- while the decode above runs on the main stream
            with torch.cuda.stream(side):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                state["bases"] = comm.allgatherv(comp, offs, concat, table, stream=side)
                e1.record()
                gather_ms.append((e0, e1))
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

    timing_local = {k: v["ms"] / max(1, v["launches"]) for k, v in timing.items()}
    codec_ms = sum(timing_local.values())
    if exchange:
        t = torch.tensor([wall, codec_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, codec_ms = float(t[0]), float(t[1])
        ok = torch.tensor([1 if roundtrip_ok else 0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        roundtrip_ok = bool(int(ok[0]))
        # the concatenated stream must hold this rank's segment at its base, and the global table its offsets
        seg_base, block_base = state["bases"]
        roundtrip_ok = roundtrip_ok and bool(torch.equal(concat[seg_base[rank]: seg_base[rank] + total], comp[:total]))
        roundtrip_ok = roundtrip_ok and bool(torch.equal(table[block_base[rank]: block_base[rank] + nblocks] - seg_base[rank], offs[:nblocks]))
        roundtrip_ok = roundtrip_ok and int(table[block_base[-1]]) == seg_base[-1]

    if rank == 0:
        workload_origin = ("the 11 Canterbury files concatenated and tiled" if args.workload == "canterbury"
                           else f"std::mt19937({seed if world == 1 else '12345 + rank'})")
        ms_step = wall * 1e3 / args.steps
        enc_ms = timing["encode"]["ms"] / max(1, timing["encode"]["launches"])
        dec_ms = timing["decode"]["ms"] / max(1, timing["decode"]["launches"])
        scan_ms = timing["scan"]["ms"] / max(1, timing["scan"]["launches"])
        scat_ms = timing["scatter"]["ms"] / max(1, timing["scatter"]["launches"])
        auto = "rcx_dec_quad_k"  # rcx_api.hip: decode_lanes()
        dec_name = {"1": "rcx_dec_adaptive_k", "4": "rcx_dec_quad_k", "8": "rcx_dec_oct_k"}.get(os.environ.get("RCX_LANES_PER_BLOCK", ""), auto)
        enc_name = {"0": "rcx_enc_adaptive_k", "1": "rcx_enc_oct_k", "2": "rcx_enc_mc_k"}.get(os.environ.get("RCX_ENC_VARIANT", ""), "rcx_enc_mc5_k")
        if coder != 0:  # (kernel names of the sibling coders, rcx_api.hip)
            enc_name = {1: "rcx_enc_static3_k", 2: "rcx_enc_rans1_k", 3: "rcx_enc_rans_k"}[coder]
            dec_name = {1: "rcx_dec_static_quad_k", 2: "rcx_dec_rans1_quad_k", 3: "rcx_dec_rans8_k"}[coder]
        dom, dom_ms = (dec_name, dec_ms) if dec_ms >= enc_ms else (enc_name, enc_ms)
        algo_bytes = (1.0 + ratio) * n  # 1 read + r write per input byte (or r read + 1 write)
        fwd_ms = timing["bwt_forward"]["ms"] / max(1, timing["bwt_forward"]["launches"])
        inv_ms = timing["bwt_inverse"]["ms"] / max(1, timing["bwt_inverse"]["launches"])
        if args.blksort and max(fwd_ms, inv_ms) > dom_ms:  # the transform reads a block and writes it (+ 2 bytes)
            dom, dom_ms = ("rcx_bwt_fwd_k", fwd_ms) if fwd_ms >= inv_ms else ("rcx_bwt_inv_k", inv_ms)
            algo_bytes = float(n + m)
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel from the PMC passes (separate rocprofv3 --pmc runs cannot happen
        # inside this process): replayed from profiles/pmc_traffic.json, and only if that file was recorded on the
        # workload this run is on -- otherwise null.
        traffic, traffic_source, ceiling = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                cfg = rec.get("_config", {})
                if (cfg.get("workload"), cfg.get("bytes"), cfg.get("block")) == (args.workload, n, block):
                    traffic = rec.get(dom, {}).get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/pmc_traffic.json ({cfg.get('recorded', 'an earlier rocprofv3 --pmc run')} on this workload), not measured by this run"
                    ceiling = issue_ceiling(rec, dom, block)
                    if ceiling:
                        ceiling["source"] = traffic_source
                else:
                    traffic_source = "none: profiles/pmc_traffic.json was recorded on another workload"
            except Exception:
                traffic = None
        line = {
            "metric": "MB/s encode+decode per GPU on 64 KiB blocks; ratio vs CPU ref",
            "value": round(world * n / 1e6 / (ms_step * 1e-3), 1),
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{n >> 20} MiB synthetic {args.workload} bytes per GPU ({workload_origin}), "
                                   f"{block >> 10} KiB blocks, encode+decode round trip resident in HBM",
                       "baseline_config": "none of BASELINE.json's configs: the reference's block sort (blksort.h) in front of the coder" if args.blksort else
                                          baseline_config(args.workload, n, block, world) if coder == 0 else
                                          f"none of BASELINE.json's configs: the {args.coder} coder (cppans.h / RangeEncoder) on that shape",
                       "coder": args.coder, "blksort": bool(args.blksort),
                       "bytes_per_gpu": n, "block": block, "blocks_per_gpu": nblocks,
                       "parallelism": f"blocks sharded over {world} GPU(s), one process per GPU"
                                      + ("; rcx_allgatherv_segments (RCCL send/recv straight into place) overlapped with the decode and with the next step's encode" if exchange else "")},
            "roundtrip_ok": roundtrip_ok, "ratio": round(ratio, 6),
            "encode_MBps": round(n / 1e6 / ((enc_ms + scan_ms + scat_ms) * 1e-3), 1),
            "decode_MBps": round(n / 1e6 / (dec_ms * 1e-3), 1),
            "kernel_ms": dict({"encode": round(enc_ms, 4), "scan": round(scan_ms, 4), "scatter": round(scat_ms, 4), "decode": round(dec_ms, 4)},
                              **({"bwt_forward": round(fwd_ms, 4), "bwt_inverse": round(inv_ms, 4)} if args.blksort else {})),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "issue_ceiling": ceiling,
                         "note": "per-block coding is a serial chain per symbol; the ceiling that actually binds is "
                                 "blocks_in_flight x clock / cycles_per_symbol (DESIGN.md), not HBM: issue_ceiling says how much of a "
                                 "wave's time along that chain goes into issuing its vector instructions"},
            "workload_gen_s": round(gen_s, 2),
        }
        if exchange and gather_ms:
            # codec-only and codec+exchange separately.  `value` above is the whole step (the
            # exchange overlaps the decode and the next encode); codec-only is the sum of the kernel times (max over ranks).
            ag_ms = sum(a.elapsed_time(b) for a, b in gather_ms) / len(gather_ms)
            line["allgatherv_ms"] = round(ag_ms, 3)
            line["codec_only_MBps"] = round(world * n / 1e6 / (codec_ms * 1e-3), 1)
            line["codec_plus_allgatherv_MBps"] = line["value"]
            line["allgatherv_in_GBps_per_gpu"] = round((world - 1) * total / 1e9 / (ag_ms * 1e-3), 2) if world > 1 else 0.0
        if world == 1 and not args.no_cpu_baseline:
            try:
                sample = host_src[: min(n, args.cpu_sample_mib << 20) // block * block or n]
                if args.blksort:  # the transform costs the CPU ~ 5 ms a block: a smaller sample (whole 32 KiB blocks)
                    sample = sample[: min(len(sample), 64 << 20)]
                cb, cpu_sizes = cpu_baseline(sample, block, coder, args.blksort)
                line["cpu_baseline"] = cb
                line["gpu_over_cpu"] = round(line["value"] / cb["value"], 1)
                # same bytes on both sides: the GPU's per-block stream sizes must be the CPU coder's (with the block
                # sort in front the sample's last coder block is cut differently: all but that one)
                k = len(cpu_sizes) - (1 if args.blksort else 0)
                gpu_sizes = (offs[1: k + 1] - offs[:k]).cpu().numpy()
                line["cpu_baseline"]["block_sizes_equal_gpu"] = bool(np.array_equal(gpu_sizes, cpu_sizes[:k].astype(np.int64)))
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": usable_cpus(), "kind": "port", "sample": f"failed: {e}"}
        if world == 1 and not args.no_end_to_end and not args.blksort:
            try:
                line["end_to_end"] = end_to_end(ctx, host_src, block, coder)
            except Exception as e:  # a report beside the number, never a reason to lose it
                line["end_to_end"] = {"encode_MBps": None, "decode_MBps": None, "what": f"failed: {e}"}
        os.write(_REAL_STDOUT, (json.dumps(line) + "\n").encode())

    ctx.close()
    if exchange:
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
