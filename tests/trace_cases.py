"""Inputs for the tests of the resumable encoder (rcx_estream_*, the facade's AdaptiveRangeEncoder<T>::encode fed in pieces):
(name, bytes, piece size, sink capacity or None).  tests/golden/make_golden_traces.py runs the real reference over them and
stores the sink's size after every call; the tests run the oracle and the GPU path over the same inputs."""
from __future__ import annotations

import numpy as np

import carry_runs
from cpprcoder_amd import workloads


def cases():
    out = []
    text = workloads.canterbury_files()["alice29.txt"]
    zipf = workloads.zipf(30000, 11).tobytes()
    uni = workloads.uniform(20000, 12345).tobytes()
    out.append(("empty", b"", 1, None))
    out.append(("one byte", b"\x00", 1, None))
    out.append(("hello world, byte by byte", b"hello world", 1, None))
    out.append(("zipf in pieces of 777", zipf, 777, None))
    out.append(("zipf in pieces of 1", zipf[:3000], 1, None))
    out.append(("uniform in pieces of 4096", uni, 4096, None))
    out.append(("alice29 in pieces of 10000", text, 10000, None))
    out.append(("ff x 5000 in pieces of 100", b"\xff" * 5000, 100, None))
    # long runs of held 0xFF bytes (tests/carry_runs.py): the sink lags by the whole run until a carry or a smaller byte ends it
    for run in (3, 40, 300):
        data = carry_runs.carry_run_block(4096, run, 5 + run)
        out.append((f"carry through a run of {run} pending bytes, pieces of 64", bytes(data), 64, None))
    # a sink that fills: in a symbol of some piece, and in the last piece's finish()
    out.append(("zipf, sink of 5000 bytes", zipf, 777, 5000))
    out.append(("uniform, sink of 16384 bytes", uni, 4096, 16384))
    out.append(("uniform, the sink fills in a symbol of the last piece", uni[:4000], 1000, 4032))
    # the sink fills in the middle of a group: the held byte and part of a pending run of 300 bytes still fit
    out.append(("the sink fills inside a pending run", bytes(carry_runs.carry_run_block(4096, 300, 305)), 64, 128))
    out.append(("the sink fills inside a pending run, later", bytes(carry_runs.carry_run_block(4096, 300, 305)), 64, 256))
    # 3013 uniform bytes make a stream of 3061 bytes: a sink of 3056 takes every byte the symbols put out and fails in
    # finish(), which the reference's encode() does not report (cpprcoder.h:716)
    out.append(("uniform, only finish() finds the sink full", workloads.uniform(6000, 777).tobytes()[:3013], 500, 3056))
    return out
