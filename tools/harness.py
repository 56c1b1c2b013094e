#!/usr/bin/env python3
"""The reference harness's protocol on the GPU path: one row per Canterbury file.

test/main.cpp:304-364 (run_adaptive) loads a file, times initialize()+encode() and
initialize()+decode() of ONE stream per file, compares the round trip and prints
|file|ratio|encode|decode|.  This does the same through the C ABI's single-stream calls
(rcx_stream_encode / rcx_stream_decode, reference-exact stream semantics).  A single stream is a
single serial chain = one GPU lane, so these times are latency, not throughput; the second table
codes the same file as 4 KiB blocks (the many-block path) for comparison.  Ratio = compressed/original
as in README.md:16-46.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from cpprcoder_amd import rcx, workloads
    files = workloads.canterbury_files()
    ctx = rcx.Context(0)
    ctx.stream_encode(b"warm up")
    # test/main.cpp:253-302 (run_rangecoder), :304-364 (run_adaptive), :366-456 (run_ans), :458-546 (run_ans_simd)
    for title, coder in (("Adaptive Range Coder", rcx.CODER_ADAPTIVE), ("Range Coder (static)", rcx.CODER_STATIC),
                         ("rANS", rcx.CODER_RANS), ("rANS SIMD (rANS::encode_simd / decode_simd)", rcx.CODER_RANS8)):
        print(f"{title}, one stream per file (rcx_stream_encode / rcx_stream_decode)")
        print("|file|bytes|compressed|ratio|encode us|decode us|round trip|")
        print("|:---|---:|---:|---:|---:|---:|:---|")
        for name in workloads.CANTERBURY_ORDER:
            data = files[name]
            t0 = time.perf_counter()
            st, rq, comp = ctx.stream_encode(data, coder=coder)
            t1 = time.perf_counter()
            st2, rq2, back = ctx.stream_decode(comp, len(data), coder=coder)
            t2 = time.perf_counter()
            ok = st == 0 and st2 == 0 and back == data
            print(f"|{name}|{len(data)}|{len(comp)}|{len(comp) / len(data):.6f}|{(t1 - t0) * 1e6:.0f}|{(t2 - t1) * 1e6:.0f}|{'ok' if ok else 'MISMATCH'}|")
        print()
    print("Same files as independent 4 KiB blocks (rcx_encode_blocks / rcx_decode_blocks, host buffers incl. PCIe copies)")
    print("|file|bytes|compressed|ratio|encode us|decode us|round trip|")
    print("|:---|---:|---:|---:|---:|---:|:---|")
    ctx.encode_blocks(np.zeros(8192, np.uint8), 4096)
    for name in workloads.CANTERBURY_ORDER:
        data = np.frombuffer(files[name], np.uint8)
        t0 = time.perf_counter()
        payload, offsets = ctx.encode_blocks(data, 4096)
        t1 = time.perf_counter()
        back = ctx.decode_blocks(payload, offsets, 4096)
        t2 = time.perf_counter()
        ok = np.array_equal(back, data)
        print(f"|{name}|{len(data)}|{len(payload)}|{len(payload) / len(data):.6f}|{(t1 - t0) * 1e6:.0f}|{(t2 - t1) * 1e6:.0f}|{'ok' if ok else 'MISMATCH'}|")
    print()
    # test/main.cpp:790-840 (run_blksort): BlkSort::encode + decode of the whole file; "ratio" there is size / encoded size
    print("BLKSORT (rcx_bwt_encode / rcx_bwt_decode, host buffers incl. PCIe copies), and the block-sorted file through the")
    print("adaptive coder in 64 KiB blocks next to the plain file (container --blksort)")
    print("|file|bytes|encoded|encode us|decode us|round trip|adaptive 64 KiB plain|after block sort|")
    print("|:---|---:|---:|---:|---:|:---|---:|---:|")
    ctx.bwt_encode(np.zeros(65536, np.uint8))
    for name in workloads.CANTERBURY_ORDER:
        data = np.frombuffer(files[name], np.uint8)
        t0 = time.perf_counter()
        enc = ctx.bwt_encode(data)
        t1 = time.perf_counter()
        back = ctx.bwt_decode(enc)
        t2 = time.perf_counter()
        ok = np.array_equal(back, data)
        plain, _ = ctx.encode_blocks(data, 65536)
        sorted_first, _ = ctx.encode_blocks(enc, 65536)
        print(f"|{name}|{len(data)}|{len(enc)}|{(t1 - t0) * 1e6:.0f}|{(t2 - t1) * 1e6:.0f}|{'ok' if ok else 'MISMATCH'}|"
              f"{len(plain) / len(data):.6f}|{len(sorted_first) / len(data):.6f}|")
    ctx.close()


if __name__ == "__main__":
    main()
