"""ctypes bindings for the test-only CPU checkers.

``oracle()``    -> oracle/liboracle.so, the plain-C restatement (always available,
                   built by ``make -C oracle`` / ``__graft_entry__.build()``).
``reference()`` -> oracle/_ref/libcpprcoder_ref.so, the unmodified reference header
                   compiled in the build container (``make -C oracle ref``); None if absent.

Both export the same entry points under the prefixes ``rco_`` / ``ref_`` so one
``Checker`` class serves both.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libcpprcoder_ref.so")
REF_ANS_SO = os.path.join(ORACLE_DIR, "_ref", "libcppans_ref.so")
REF_BWT_SO = os.path.join(ORACLE_DIR, "_ref", "libblksort_ref.so")

BWT_BLOCK, BWT_ENCODED = 32768, 32770  # blksort.h:82, :85

CODER_ADAPTIVE, CODER_STATIC, CODER_RANS, CODER_RANS8 = 0, 1, 2, 3

SUCCESS, PENDING, ERROR = 0, 1, -1


class _Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("request_size", C.c_uint32)]


def _u8(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class Checker:
    def __init__(self, path: str, prefix: str, kind: str, ans_path: str | None = None, bwt_path: str | None = None):
        self.lib = C.CDLL(path)
        # block sort (blksort.h): likewise
        self.bwt = C.CDLL(bwt_path) if bwt_path else (self.lib if prefix == "rco_" else None)
        # rANS (cppans.h): in liboracle.so itself for the restatement, in its own library for the reference build
        self.ans = C.CDLL(ans_path) if ans_path else (self.lib if prefix == "rco_" else None)
        self.prefix = prefix
        self.kind = kind  # "port" (restatement) or "reference"
        p, L = prefix, self.lib
        u8p, u64p, u32p, i32p = C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p
        self._aenc = getattr(L, p + "adaptive_encode")
        self._aenc.restype, self._aenc.argtypes = _Result, [u8p, C.c_uint32, u8p, C.c_uint64, u64p]
        self._adec = getattr(L, p + "adaptive_decode")
        self._adec.restype, self._adec.argtypes = _Result, [u8p, C.c_uint64, u8p, C.c_uint64, u64p]
        self._aencc = getattr(L, p + "adaptive_encode_chunked")
        self._aencc.restype, self._aencc.argtypes = _Result, [u8p, C.c_uint32, C.c_uint32, u8p, C.c_uint64, u64p]
        self._aenct = getattr(L, p + "adaptive_encode_trace", None)
        if self._aenct is not None:
            self._aenct.restype = C.c_uint32
            self._aenct.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p, C.c_uint64, u64p, C.c_void_p, C.c_uint32, C.POINTER(_Result)]
        self._adecc = getattr(L, p + "adaptive_decode_chunked")
        self._adecc.restype, self._adecc.argtypes = _Result, [u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint64, u64p]
        self._senc = getattr(L, p + "static_encode")
        self._senc.restype, self._senc.argtypes = C.c_int, [u8p, C.c_uint32, u8p, C.c_uint64, u64p]
        self._sdec = getattr(L, p + "static_decode")
        self._sdec.restype, self._sdec.argtypes = C.c_int, [u8p, C.c_uint32, u8p, C.c_uint64, u64p]
        self._ebr = getattr(L, p + "encode_block_range")
        self._ebr.restype = C.c_int
        self._ebr.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, u8p, C.c_uint64, u32p, C.c_int]
        self._dbr = getattr(L, p + "decode_block_range")
        self._dbr.restype = C.c_int
        self._dbr.argtypes = [u8p, C.c_uint64, u32p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, u8p, C.c_int]
        self._probe = getattr(L, p + "model_probe")
        self._probe.restype = None
        self._probe.argtypes = [u8p, C.c_uint64, u32p, u32p, u32p, u32p, C.c_uint32, u32p, u8p]
        self._script = getattr(L, p + "stream_script")
        self._script.restype, self._script.argtypes = C.c_int, [i32p, C.c_int, i32p]
        if self.ans is not None:
            A = self.ans
            self._rebr = getattr(A, p + "rans_encode_block_range")
            self._rebr.restype = C.c_int
            self._rebr.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, u8p, C.c_uint64, u32p, C.c_int]
            self._rdbr = getattr(A, p + "rans_decode_block_range")
            self._rdbr.restype = C.c_int
            self._rdbr.argtypes = [u8p, C.c_uint64, u32p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, u8p, C.c_int]

        if self.bwt is not None:
            B = self.bwt
            for name in ("bwt_encode_bound", "bwt_decode_bound", "bwt_decoded_size"):
                fn = getattr(B, p + name)
                fn.restype, fn.argtypes = C.c_uint64, [C.c_uint64]
            for name in ("bwt_encode", "bwt_decode"):
                fn = getattr(B, p + name)
                fn.restype, fn.argtypes = C.c_int, [u8p, C.c_uint64, u8p]

    # ---- block sort (blksort.h) ---------------------------------------------
    def bwt_encode_bound(self, n: int) -> int:
        return int(getattr(self.bwt, self.prefix + "bwt_encode_bound")(n))

    def bwt_decode_bound(self, n: int) -> int:
        return int(getattr(self.bwt, self.prefix + "bwt_decode_bound")(n))

    def bwt_decoded_size(self, n: int) -> int:
        return int(getattr(self.bwt, self.prefix + "bwt_decoded_size")(n))

    def bwt_encode(self, data, threads: int = 1) -> np.ndarray:
        """BlkSort::encode: whole 32 KiB blocks -> 32770 bytes each, the rest copied."""
        src = _u8(data)
        out = np.zeros(self.bwt_encode_bound(len(src)), dtype=np.uint8)
        self._bwt_run("bwt_encode", src, out, BWT_BLOCK, BWT_ENCODED, threads)
        return out

    def bwt_decode(self, enc, threads: int = 1) -> np.ndarray:
        src = _u8(enc)
        out = np.zeros(self.bwt_decoded_size(len(src)), dtype=np.uint8)
        self._bwt_run("bwt_decode", src, out, BWT_ENCODED, BWT_BLOCK, threads)
        return out

    def _bwt_run(self, name, src, out, unit_in, unit_out, threads):
        fn = getattr(self.bwt, self.prefix + name)
        blocks = len(src) // unit_in

        def work(first, last):  # whole blocks [first, last); the last range also takes the copied tail
            n = (len(src) - first * unit_in) if last == blocks else (last - first) * unit_in
            if fn(src.ctypes.data + first * unit_in, n, out.ctypes.data + first * unit_out) != 0:
                raise RuntimeError(f"{self.prefix}{name} failed")

        if blocks == 0:
            work(0, 0)
        else:
            self._fan_out(work, blocks, threads)

    # ---- one-shot streams -------------------------------------------------
    @staticmethod
    def default_capacity(n: int) -> int:
        return n + n // 32 + 1024 + (n // 8 if n > (1 << 20) else 0)

    def _oneshot(self, fn, data, sink_capacity, *mid):
        src = _u8(data)
        cap = self.default_capacity(len(src)) if sink_capacity is None else int(sink_capacity)
        out = np.zeros(max(cap, 16) + 32, dtype=np.uint8)
        size = C.c_uint64()
        res = fn(src.ctypes.data, len(src), *mid, out.ctypes.data, cap, C.byref(size))
        return res, bytes(out[: min(size.value, cap)]), size.value

    def adaptive_encode(self, data, sink_capacity=None):
        """-> ((status, request_size), stream bytes, stream size)"""
        res, out, size = self._oneshot(self._aenc, data, sink_capacity)
        return (res.status, res.request_size), out, size

    def adaptive_encode_chunked(self, data, piece, sink_capacity=None):
        res, out, size = self._oneshot(self._aencc, data, sink_capacity, piece)
        return (res.status, res.request_size), out, size

    def adaptive_encode_trace(self, data, piece, sink_capacity=None):
        """encode() in pieces of `piece` bytes -> ((status, request_size) of the last call, the sink's bytes, the sink's size
        after initialize() and after every encode() call)."""
        src = _u8(data)
        cap = self.default_capacity(len(src)) if sink_capacity is None else int(sink_capacity)
        out = np.zeros(max(cap, 16) + 32, dtype=np.uint8)
        calls = (len(src) + piece - 1) // piece + 3
        sizes = np.zeros(calls, dtype=np.uint32)
        size, last = C.c_uint64(), _Result()
        got = self._aenct(src.ctypes.data, len(src), piece, out.ctypes.data, cap, C.byref(size), sizes.ctypes.data, calls, C.byref(last))
        return (last.status, last.request_size), bytes(out[: min(size.value, cap)]), [int(x) for x in sizes[:got]]

    def adaptive_decode(self, comp, sink_capacity):
        res, out, size = self._oneshot(self._adec, comp, sink_capacity)
        return (res.status, res.request_size), out, size

    def adaptive_decode_chunked(self, comp, piece, sink_capacity):
        res, out, size = self._oneshot(self._adecc, comp, sink_capacity, piece)
        return (res.status, res.request_size), out, size

    def static_encode(self, data, sink_capacity=None):
        src = _u8(data)
        cap = (len(src) + len(src) // 32 + 2048) if sink_capacity is None else int(sink_capacity)
        ok, out, size = self._oneshot(self._senc, src, cap)
        return bool(ok), out, size

    def static_decode(self, comp, sink_capacity):
        ok, out, size = self._oneshot(self._sdec, comp, sink_capacity)
        return bool(ok), out, size

    # ---- rANS (cppans.h), one stream ---------------------------------------
    def rans_encode(self, data, simd: bool = False) -> bytes:
        """rANS::encode / encode_simd on one buffer -> the stream (the last `size` bytes of the reference's destination)."""
        src = _u8(data)
        slots, sizes = self.encode_blocks(src, max(len(src), 1), coder=CODER_RANS8 if simd else CODER_RANS)
        return bytes(slots[0, : int(sizes[0])])

    def rans_decode(self, comp, n: int, simd: bool = False):
        """-> (ok, bytes)"""
        c = _u8(comp)
        slots = np.zeros((1, len(c) + 64), dtype=np.uint8)
        slots[0, : len(c)] = c
        out, ok = self.decode_blocks(slots, np.array([len(c)], np.uint32), max(n, 1), n, coder=CODER_RANS8 if simd else CODER_RANS)
        return ok, out.tobytes()

    # ---- many blocks ------------------------------------------------------
    @staticmethod
    def block_bound(block: int, coder: int = 0) -> int:
        if coder >= CODER_RANS:
            # cppans.h:492-495 (2n + 1032) + 64: encode_simd's eight flushed states, and its 2 bytes per symbol for a
            # block of one repeated byte (the renormalisation test of cppans.h:357 wraps to "always" for freq = 4096)
            return (2 * block + 1032 + 64 + 15) & ~15
        b = block + block // 32 + 2048 + (block // 8 if block > (1 << 20) else 0)
        return (b + 15) & ~15

    def encode_blocks(self, data, block: int, coder: int = 0, threads: int = 1):
        """-> (slots[nblocks, slot] u8, sizes[nblocks] u32).  Threads split the block range
        (ctypes drops the GIL for the duration of each call)."""
        src = _u8(data)
        n = len(src)
        nblocks = (n + block - 1) // block
        slot = self.block_bound(block, coder)
        slots = np.zeros((nblocks, slot), dtype=np.uint8)
        sizes = np.zeros(nblocks, dtype=np.uint32)
        oks = []

        def work(first, last):
            if coder >= CODER_RANS:
                oks.append(self._rebr(src.ctypes.data, n, block, first, last, slots.ctypes.data, slot, sizes.ctypes.data, coder - CODER_RANS))
            else:
                oks.append(self._ebr(src.ctypes.data, n, block, first, last, slots.ctypes.data, slot, sizes.ctypes.data, coder))

        self._fan_out(work, nblocks, threads)
        if not all(oks):
            raise RuntimeError(f"{self.prefix}encode_block_range reported a failure")
        return slots, sizes

    def decode_blocks(self, slots, sizes, block: int, n: int, coder: int = 0, threads: int = 1):
        slots = np.ascontiguousarray(slots, dtype=np.uint8)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        nblocks, slot = slots.shape
        out = np.zeros(n, dtype=np.uint8)
        oks = []

        def work(first, last):
            if coder >= CODER_RANS:
                oks.append(self._rdbr(slots.ctypes.data, slot, sizes.ctypes.data, block, n, first, last, out.ctypes.data, coder - CODER_RANS))
            else:
                oks.append(self._dbr(slots.ctypes.data, slot, sizes.ctypes.data, block, n, first, last, out.ctypes.data, coder))

        self._fan_out(work, nblocks, threads)
        return out, all(oks)

    @staticmethod
    def _fan_out(work, nblocks, threads):
        threads = max(1, min(threads, nblocks))
        if threads == 1:
            work(0, nblocks)
            return
        cuts = [nblocks * i // threads for i in range(threads + 1)]
        pool = [threading.Thread(target=work, args=(cuts[i], cuts[i + 1])) for i in range(threads)]
        for t in pool:
            t.start()
        for t in pool:
            t.join()

    @staticmethod
    def compact(slots, sizes):
        """Concatenate the per-block streams -> (payload bytes, offsets[nblocks+1] u64)."""
        sizes = np.asarray(sizes, dtype=np.uint64)
        offsets = np.zeros(len(sizes) + 1, dtype=np.uint64)
        np.cumsum(sizes, out=offsets[1:])
        payload = np.concatenate([slots[b, : int(sizes[b])] for b in range(len(sizes))]) if len(sizes) else np.zeros(0, np.uint8)
        return payload, offsets

    # ---- probes -----------------------------------------------------------
    def model_probe(self, symbols, targets):
        syms = _u8(symbols)
        tg = np.ascontiguousarray(targets, dtype=np.uint32)
        total = np.zeros(1, dtype=np.uint32)
        freq = np.zeros(256, dtype=np.uint32)
        cum = np.zeros(256, dtype=np.uint32)
        fcount = np.zeros(len(tg), dtype=np.uint32)
        fcode = np.zeros(len(tg), dtype=np.uint8)
        self._probe(syms.ctypes.data, len(syms), total.ctypes.data, freq.ctypes.data, cum.ctypes.data,
                    tg.ctypes.data, len(tg), fcount.ctypes.data, fcode.ctypes.data)
        return int(total[0]), freq, cum, fcount, fcode

    def stream_script(self, ops):
        """ops: [(opcode, arg), ...] -> [(ret, capacity, size), ...] (see ref_shim.cpp)."""
        flat = np.asarray(ops, dtype=np.int32).reshape(-1)
        out = np.zeros(3 * len(ops), dtype=np.int32)
        w = self._script(flat.ctypes.data, len(ops), out.ctypes.data)
        assert w == 3 * len(ops)
        return [tuple(int(v) for v in out[3 * i: 3 * i + 3]) for i in range(len(ops))]


def build_oracle(force: bool = False) -> None:
    """Compile the C restatement (and the reference build when /root/reference exists)."""
    sources = [os.path.join(ORACLE_DIR, f) for f in ("rc_oracle.c", "rans_oracle.c", "bwt_oracle.c")]
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in sources):
        subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, capture_output=True)
    ref_header = os.environ.get("RCX_REFERENCE", "/root/reference")
    if os.path.exists(os.path.join(ref_header, "cpprcoder.h")):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref", f"REFERENCE={ref_header}"], check=True, capture_output=True)


_cache: dict[str, Checker | None] = {}


def oracle() -> Checker:
    if "o" not in _cache:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _cache["o"] = Checker(ORACLE_SO, "rco_", "port")
    return _cache["o"]


def reference() -> Checker | None:
    if "r" not in _cache:
        _cache["r"] = (Checker(REF_SO, "ref_", "reference", REF_ANS_SO if os.path.exists(REF_ANS_SO) else None,
                               REF_BWT_SO if os.path.exists(REF_BWT_SO) else None)
                       if os.path.exists(REF_SO) else None)
    return _cache["r"]


def fnv1a64(data) -> int:
    lib = oracle().lib
    lib.rco_fnv1a64.restype, lib.rco_fnv1a64.argtypes = C.c_uint64, [C.c_void_p, C.c_uint64]
    a = _u8(data)
    return int(lib.rco_fnv1a64(a.ctypes.data, len(a)))
