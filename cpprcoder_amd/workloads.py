"""Seeded synthetic byte buffers for the parity tests and bench.py.

These are the inputs SURVEY.md section 8(d) defines for BASELINE.json's configs.
Nothing here touches the GPU or the oracle; it only makes bytes.

* uniform  -- ``std::mt19937 mt(seed)``, each 32-bit draw gives 4 bytes little-endian
              (numpy's legacy ``RandomState(seed)`` is the same generator and
              ``.bytes()`` has exactly that byte order; checked against libstdc++).
* zipf     -- symbol k in [0,255] with P(k) ~ 1/(k+1), inverse CDF on
              ``u = (mt() + 0.5) / 2**32`` with ``mt19937(seed)``.
* canterbury_tiled -- the 11 Canterbury files in the order of the reference
              harness (test/main.cpp:1248-1258) concatenated and tiled cyclically.
              The corpus is reference-shipped *test data* (test/cantrbry.tar.bz2),
              committed as a fixture under tests/golden/.
"""
from __future__ import annotations

import io
import os
import tarfile

import numpy as np

CANTERBURY_ORDER = (
    "alice29.txt", "asyoulik.txt", "cp.html", "fields.c", "grammar.lsp", "kennedy.xls",
    "lcet10.txt", "plrabn12.txt", "ptt5", "sum", "xargs.1",
)  # reference: test/main.cpp:1248-1258

_GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CANTERBURY_ARCHIVE = os.path.join(_GOLDEN_DIR, "cantrbry.tar.bz2")


def _mt19937_draws(seed: int):
    """std::mt19937(seed)'s 32-bit draws, a chunk at a time (numpy's MT19937 with the legacy init_genrand seeding is
    that generator; about 7x faster than RandomState.bytes / .randint, same values)."""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)

    def take(count: int) -> np.ndarray:
        return bg.random_raw(count).astype(np.uint32)

    return take


def uniform(n: int, seed: int = 12345) -> np.ndarray:
    """n uniform random bytes."""
    words = (n + 3) // 4
    out = np.empty(words, dtype=np.uint32)
    take = _mt19937_draws(seed)
    step = 1 << 22
    for at in range(0, words, step):
        m = min(step, words - at)
        out[at:at + m] = take(m)
    return out.view(np.uint8)[:n].copy()


_ZIPF_TABLES = None


def _zipf_tables():
    """Integer form of the inverse CDF: symbol = number of k with thr[k] <= draw, where thr[k] is the smallest draw d
    with (d + 0.5) / 2**32 > cdf[k] -- exactly what np.searchsorted(cdf, u, side="left") counts.  Plus a lookup on the
    top 16 bits of the draw that settles all but the ~255 boundary cells."""
    global _ZIPF_TABLES
    if _ZIPF_TABLES is None:
        weights = 1.0 / np.arange(1, 257, dtype=np.float64)
        cdf = np.cumsum(weights)
        cdf /= cdf[-1]
        thr = np.empty(256, dtype=np.uint64)
        for k in range(256):
            lo, hi = 0, 1 << 32  # smallest d in [0, 2^32] with (d + 0.5) / 2^32 > cdf[k]
            while lo < hi:
                mid = (lo + hi) // 2
                if (np.float64(mid) + 0.5) / 4294967296.0 > cdf[k]:
                    hi = mid
                else:
                    lo = mid + 1
            thr[k] = lo
        cells = np.arange(1 << 16, dtype=np.uint64) << np.uint64(16)
        first = np.searchsorted(thr, cells, side="right")                    # symbol of the cell's first draw
        last = np.searchsorted(thr, cells + np.uint64(0xFFFF), side="right")  # ... and of its last
        _ZIPF_TABLES = (thr, np.minimum(first, 255).astype(np.uint8), first != last)
    return _ZIPF_TABLES


def zipf(n: int, seed: int = 12345) -> np.ndarray:
    """n Zipf(s=1) bytes over 256 symbols."""
    thr, cell_sym, cell_split = _zipf_tables()
    out = np.empty(n, dtype=np.uint8)
    take = _mt19937_draws(seed)
    step = 1 << 22
    for at in range(0, n, step):
        m = min(step, n - at)
        draws = take(m)
        top = draws >> np.uint32(16)
        sym = cell_sym[top]
        hard = np.flatnonzero(cell_split[top])
        if len(hard):
            sym[hard] = np.minimum(np.searchsorted(thr, draws[hard].astype(np.uint64), side="right"), 255).astype(np.uint8)
        out[at:at + m] = sym
    return out


def canterbury_files(archive: str = CANTERBURY_ARCHIVE) -> dict[str, bytes]:
    """The corpus files by name (read from the committed archive, nothing is executed)."""
    files: dict[str, bytes] = {}
    with tarfile.open(archive, "r:bz2") as tar:
        for member in tar.getmembers():
            if member.isfile():
                files[os.path.basename(member.name)] = tar.extractfile(member).read()
    return files


def canterbury_concat(archive: str = CANTERBURY_ARCHIVE) -> np.ndarray:
    files = canterbury_files(archive)
    return np.frombuffer(b"".join(files[name] for name in CANTERBURY_ORDER), dtype=np.uint8)


def canterbury_tiled(n: int, archive: str = CANTERBURY_ARCHIVE) -> np.ndarray:
    """buf[i] = cat[i mod len(cat)]."""
    cat = canterbury_concat(archive)
    reps = (n + len(cat) - 1) // len(cat)
    return np.tile(cat, reps)[:n].copy()


def runs(n: int, seed: int = 7) -> np.ndarray:
    """Long single-symbol runs with rare breaks: drives the encoder's carry /
    pending-0xFF branches and highly skewed tables."""
    rs = np.random.RandomState(seed)
    out = np.empty(n, dtype=np.uint8)
    at = 0
    while at < n:
        length = int(rs.randint(1, 4096))
        out[at:at + length] = rs.randint(0, 256)
        at += length
    return out


def by_name(name: str, n: int, seed: int = 12345) -> np.ndarray:
    if name == "uniform":
        return uniform(n, seed)
    if name == "zipf":
        return zipf(n, seed)
    if name == "canterbury":
        return canterbury_tiled(n)
    if name == "runs":
        return runs(n, seed)
    raise ValueError(f"unknown workload {name!r}")
