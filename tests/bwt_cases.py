"""Inputs for the block-sort (blksort.h) parity tests, built from integer arithmetic only so that the same bytes come
out on every machine and numpy version.  tests/golden/make_golden_bwt.py runs the REAL reference over them and stores
what it produced (tests/golden/bwt.json); the tests run the oracle and the GPU path over the same inputs.

A case is a name -> bytes.  The block size is fixed by the reference (32768, blksort.h:82); what exercises the code is
the content: ordinary data, long repeats (deep comparisons), and periodic blocks, where the rotations tie and the row
index the reference stores depends on the moves of its unstable sort (oracle/bwt_oracle.c).
"""
from __future__ import annotations

import numpy as np

BLOCK = 32768
ENCODED = BLOCK + 2


def mix(n: int, seed: int) -> np.ndarray:
    """n pseudo-random bytes: splitmix64 of (seed, index), top byte."""
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)) * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(31)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(29)
    return (x >> np.uint64(56)).astype(np.uint8)


def _tile(word: np.ndarray, n: int = BLOCK) -> np.ndarray:
    return np.resize(word, n).astype(np.uint8)


def periodic(p: int, seed: int, alphabet: int = 256) -> np.ndarray:
    """A block of period p.  The word is random, so p is its primitive period with near certainty; seeds below were
    chosen (and are checked in the generator) so that it is."""
    return _tile((mix(p, seed).astype(np.uint32) % alphabet).astype(np.uint8))


def cases() -> dict[str, np.ndarray]:
    c: dict[str, np.ndarray] = {}
    c["random 2 blocks + tail"] = mix(2 * BLOCK + 1234, 1)
    c["tail only"] = mix(1000, 2)
    c["empty"] = np.zeros(0, np.uint8)
    c["four letters"] = (mix(BLOCK, 3) % 4 + 97).astype(np.uint8)
    c["two letters"] = (mix(BLOCK, 4) % 2 + 97).astype(np.uint8)
    c["runs of 64"] = np.repeat(mix(BLOCK // 64, 5), 64)
    c["repeat of 5000"] = _tile(mix(5000, 6))            # long common prefixes, not periodic in 32768
    c["repeat of 12288 + 1"] = np.concatenate([_tile(mix(12288, 7))[:-1], np.array([0x5A], np.uint8)])
    c["zeros"] = np.zeros(BLOCK, np.uint8)               # every rotation ties
    c["ones then a two"] = np.concatenate([np.ones(BLOCK - 1, np.uint8), np.array([2], np.uint8)])  # almost periodic
    c["ab..ab then aa"] = np.concatenate([_tile(np.frombuffer(b"ab", np.uint8))[:-2], np.frombuffer(b"aa", np.uint8)])
    for p, seed, alphabet in ((2, 11, 256), (4, 12, 256), (4, 13, 3), (16, 14, 256), (64, 15, 3), (256, 16, 256),
                              (1024, 17, 4), (4096, 18, 256), (16384, 19, 256), (16384, 20, 2)):
        c[f"period {p} (seed {seed}, {alphabet} symbols)"] = periodic(p, seed, alphabet)
    # unbalanced partitions: the reference's sort runs out of its 11 levels and falls back to heapsort (blksort.h:284-287)
    c["period 8192, skewed bytes"] = _tile(np.minimum(mix(8192, 100), 40).astype(np.uint8))
    c["period 4096, ramp"] = _tile((np.arange(4096) // 16).astype(np.uint8))
    c["period 2048, sorted"] = _tile(np.sort(mix(2048, 6)))
    # a period that is one long run and a single other byte: every partition pass of the reference's sort peels off two
    # rows, so its replay makes thousands of passes over thousands of rows (the replay's worst case, csrc/rcx_bwt_tie.hpp)
    c["period 16384, a run then b"] = _tile(np.concatenate([np.full(16383, 97, np.uint8), np.array([98], np.uint8)]))
    c["period 4096, a run then b"] = _tile(np.concatenate([np.full(4095, 97, np.uint8), np.array([98], np.uint8)]))
    c["period 8192, b then a run"] = _tile(np.concatenate([np.array([98], np.uint8), np.full(8191, 97, np.uint8)]))
    c["skewed bytes"] = np.minimum(mix(BLOCK, 101), 24).astype(np.uint8)   # not periodic, heapsort all the same
    # long runs of one byte (1 .. 4096 long): on the GPU such blocks start from run keys, not from two bytes
    c["long runs, two blocks"] = np.repeat(mix(64, 23), 1 + (mix(64, 24).astype(np.uint32) * 16 + mix(64, 25) % 16))[: 2 * BLOCK]
    c["three blocks: periodic, random, zeros"] = np.concatenate([periodic(8, 21), mix(BLOCK, 22), np.zeros(BLOCK + 77, np.uint8)])
    return c


# the subset the CPU suite runs the (slow on ties: ~1.5 s a block, like the reference) oracle over
CPU_SUBSET = ("random 2 blocks + tail", "tail only", "empty", "four letters", "runs of 64", "repeat of 5000", "zeros",
              "ones then a two", "period 2 (seed 11, 256 symbols)", "period 4 (seed 13, 3 symbols)",
              "period 256 (seed 16, 256 symbols)", "period 16384 (seed 20, 2 symbols)")


def rows(enc: np.ndarray) -> list[int]:
    """The row index stored behind every whole block of an encoded buffer."""
    nb = len(enc) // ENCODED
    return [int(enc[b * ENCODED + BLOCK]) | int(enc[b * ENCODED + BLOCK + 1]) << 8 for b in range(nb)]


def primitive_period(block: np.ndarray) -> int:
    n = len(block)
    p = 1
    while p < n:
        if n % p == 0 and np.array_equal(block, np.roll(block, -p)):
            return p
        p *= 2
    return n
