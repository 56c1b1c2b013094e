"""ctypes binding of the C ABI in include/rcx.h (cpprcoder_amd/librcx.so).

This is host plumbing for the tests and bench.py: it hands device pointers (torch tensors are
used only as HBM allocations) and a HIP stream to the native library.  There is no CPU
fallback: if librcx.so is missing, or there is no GPU when a context is created, it raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCX_LIBRARY") or os.path.join(HERE, "librcx.so")  # RCX_LIBRARY: a diagnostic (stamped) build

OK, PENDING, ERROR, E_ARG, E_CAPACITY, E_CORRUPT, E_HIP, E_NOMEM, E_COMM = 0, 1, -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128
CODER_ADAPTIVE, CODER_STATIC, CODER_RANS, CODER_RANS8 = 0, 1, 2, 3
T_ENCODE, T_SCAN, T_SCATTER, T_DECODE, T_BWT_FORWARD, T_BWT_INVERSE, T_COUNT = 0, 1, 2, 3, 4, 5, 6
BWT_BLOCK, BWT_ENCODED = 32768, 32770  # blksort.h:82, :85
MIN_BLOCK, MAX_BLOCK, MAX_STREAM = 16, (1 << 24) - 256, 0x7FFFFFFF

# every symbol include/rcx.h declares (tests check that the library exports all of them)
EXPORTS = (
    "rcx_version", "rcx_status_string", "rcx_ctx_create", "rcx_ctx_destroy", "rcx_ctx_reserve", "rcx_ctx_sync_status",
    "rcx_block_count", "rcx_block_bound", "rcx_encode_bound", "rcx_encode_blocks_device", "rcx_decode_blocks_device",
    "rcx_encode_blocks", "rcx_decode_blocks", "rcx_stream_encode", "rcx_stream_decode", "rcx_ctx_set_timing",
    "rcx_ctx_get_timing", "rcx_ctx_last_redo",
    "rcx_block_bound_for", "rcx_encode_bound_for", "rcx_ctx_reserve_for",
    "rcx_dstream_create", "rcx_dstream_destroy", "rcx_dstream_decode",
    "rcx_estream_create", "rcx_estream_destroy", "rcx_estream_encode", "rcx_estream_rewind",
    "rcx_comm_unique_id", "rcx_comm_create", "rcx_comm_destroy", "rcx_comm_rank", "rcx_comm_size", "rcx_exchange_plan",
    "rcx_allgatherv_segments",
    "rcx_bwt_encode_bound", "rcx_bwt_decode_bound", "rcx_bwt_decoded_size", "rcx_bwt_reserve", "rcx_bwt_encode_device",
    "rcx_bwt_decode_device", "rcx_bwt_encode", "rcx_bwt_decode", "rcx_bwt_last_ties",
)


class RcxError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        super().__init__(f"{where}: rcx status {status} ({status_string(status)})")


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m cpprcoder_amd.build` "
                              "(there is no CPU fallback for the HIP path)")
        # One HIP runtime per process: PyTorch's wheel carries its own libamdhip64, librcx.so asks for the same soname.
        # Whichever is loaded first serves both -- but if librcx.so comes first (the system's copy) and torch later
        # brings its own, the process has two runtimes and rcx_ctx_create fails with RCX_E_HIP.  So where torch is
        # installed it is imported first.  (A C/C++ caller without torch has only the system runtime: nothing to do.)
        if not os.environ.get("RCX_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.rcx_version.restype = i32
        L.rcx_status_string.restype, L.rcx_status_string.argtypes = C.c_char_p, [i32]
        L.rcx_ctx_create.restype, L.rcx_ctx_create.argtypes = i32, [i32, C.POINTER(vp)]
        L.rcx_ctx_destroy.restype, L.rcx_ctx_destroy.argtypes = None, [vp]
        L.rcx_ctx_reserve.restype, L.rcx_ctx_reserve.argtypes = i32, [vp, u64, u32]
        L.rcx_ctx_sync_status.restype, L.rcx_ctx_sync_status.argtypes = i32, [vp, vp, C.POINTER(u64)]
        L.rcx_block_count.restype, L.rcx_block_count.argtypes = u64, [u64, u32]
        L.rcx_block_bound.restype, L.rcx_block_bound.argtypes = u64, [u32]
        L.rcx_encode_bound.restype, L.rcx_encode_bound.argtypes = u64, [u64, u32]
        L.rcx_block_bound_for.restype, L.rcx_block_bound_for.argtypes = u64, [i32, u32]
        L.rcx_encode_bound_for.restype, L.rcx_encode_bound_for.argtypes = u64, [i32, u64, u32]
        L.rcx_ctx_reserve_for.restype, L.rcx_ctx_reserve_for.argtypes = i32, [vp, i32, u64, u32]
        L.rcx_encode_blocks_device.restype = i32
        L.rcx_encode_blocks_device.argtypes = [vp, i32, vp, u64, u32, vp, u64, vp, vp]
        L.rcx_decode_blocks_device.restype = i32
        L.rcx_decode_blocks_device.argtypes = [vp, i32, vp, u64, vp, u64, u32, u64, vp, vp]
        L.rcx_encode_blocks.restype = i32
        L.rcx_encode_blocks.argtypes = [vp, i32, vp, u64, u32, vp, u64, C.POINTER(u64), vp]
        L.rcx_decode_blocks.restype = i32
        L.rcx_decode_blocks.argtypes = [vp, i32, vp, u64, vp, u64, u32, vp, u64, C.POINTER(u64)]
        L.rcx_stream_encode.restype = i32
        L.rcx_stream_encode.argtypes = [vp, i32, vp, u32, vp, u64, u64, C.POINTER(u64), C.POINTER(u32)]
        L.rcx_stream_decode.restype = i32
        L.rcx_stream_decode.argtypes = [vp, i32, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u32)]
        L.rcx_ctx_set_timing.restype, L.rcx_ctx_set_timing.argtypes = i32, [vp, i32]
        L.rcx_ctx_get_timing.restype, L.rcx_ctx_get_timing.argtypes = i32, [vp, C.POINTER(C.c_double), C.POINTER(u64), i32]
        L.rcx_ctx_last_redo.restype, L.rcx_ctx_last_redo.argtypes = i32, [vp, u64, C.POINTER(u64)]
        L.rcx_dstream_create.restype, L.rcx_dstream_create.argtypes = i32, [vp, C.POINTER(vp)]
        L.rcx_dstream_destroy.restype, L.rcx_dstream_destroy.argtypes = None, [vp]
        L.rcx_dstream_decode.restype = i32
        L.rcx_dstream_decode.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u32)]
        L.rcx_estream_create.restype, L.rcx_estream_create.argtypes = i32, [vp, u32, C.POINTER(vp)]
        L.rcx_estream_destroy.restype, L.rcx_estream_destroy.argtypes = None, [vp]
        L.rcx_estream_encode.restype = i32
        L.rcx_estream_encode.argtypes = [vp, vp, u64, vp, u64, u64, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32)]
        L.rcx_estream_rewind.restype, L.rcx_estream_rewind.argtypes = i32, [vp]
        L.rcx_comm_unique_id.restype, L.rcx_comm_unique_id.argtypes = i32, [vp]
        L.rcx_comm_create.restype, L.rcx_comm_create.argtypes = i32, [i32, vp, i32, i32, C.POINTER(vp)]
        L.rcx_comm_destroy.restype, L.rcx_comm_destroy.argtypes = None, [vp]
        L.rcx_comm_rank.restype, L.rcx_comm_rank.argtypes = i32, [vp]
        L.rcx_comm_size.restype, L.rcx_comm_size.argtypes = i32, [vp]
        L.rcx_exchange_plan.restype, L.rcx_exchange_plan.argtypes = i32, [vp, vp, i32, vp, vp]
        L.rcx_allgatherv_segments.restype = i32
        L.rcx_allgatherv_segments.argtypes = [vp, vp, vp, u64, vp, u64, vp, u64, vp, vp, vp]
        for name in ("rcx_bwt_encode_bound", "rcx_bwt_decode_bound", "rcx_bwt_decoded_size"):
            getattr(L, name).restype, getattr(L, name).argtypes = u64, [u64]
        L.rcx_bwt_reserve.restype, L.rcx_bwt_reserve.argtypes = i32, [vp, u64]
        for name in ("rcx_bwt_encode_device", "rcx_bwt_decode_device"):
            getattr(L, name).restype, getattr(L, name).argtypes = i32, [vp, vp, u64, vp, u64, vp]
        for name in ("rcx_bwt_encode", "rcx_bwt_decode"):
            getattr(L, name).restype, getattr(L, name).argtypes = i32, [vp, vp, u64, vp, u64, C.POINTER(u64)]
        L.rcx_bwt_last_ties.restype, L.rcx_bwt_last_ties.argtypes = i32, [vp, C.POINTER(u64)]
        _lib = L
    return _lib


def status_string(status: int) -> str:
    return lib().rcx_status_string(status).decode()


def block_count(n: int, block: int) -> int:
    return int(lib().rcx_block_count(n, block))


def block_bound(block: int, coder: int = CODER_ADAPTIVE) -> int:
    return int(lib().rcx_block_bound_for(coder, block))


def encode_bound(n: int, block: int, coder: int = CODER_ADAPTIVE) -> int:
    return int(lib().rcx_encode_bound_for(coder, n, block))


def bwt_encode_bound(n: int) -> int:
    return int(lib().rcx_bwt_encode_bound(n))


def bwt_decoded_size(n: int) -> int:
    return int(lib().rcx_bwt_decoded_size(n))


def _check(status: int, where: str) -> None:
    if status != OK:
        raise RcxError(status, where)


def _np_u8(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class Context:
    """One rcx_ctx (device scratch of one GPU).  Single-threaded, like a reference coder object."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().rcx_ctx_create(device, C.byref(self._h)), "rcx_ctx_create")
        self.device = device

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().rcx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, n: int, block: int, coder: int = CODER_ADAPTIVE) -> None:
        _check(lib().rcx_ctx_reserve_for(self._h, coder, n, block), "rcx_ctx_reserve_for")

    # ---- device pointers (torch tensors as HBM) ---------------------------
    @staticmethod
    def _stream_handle(stream) -> int:
        if stream is None:
            import torch
            return torch.cuda.current_stream().cuda_stream
        return getattr(stream, "cuda_stream", stream)

    def encode_blocks_device(self, src, block: int, dst, offsets, coder: int = CODER_ADAPTIVE, stream=None) -> None:
        """src: uint8 cuda tensor; dst: uint8 cuda tensor (>= encode_bound); offsets: int64 cuda tensor [nblocks+1]."""
        n = src.numel()
        if MIN_BLOCK <= block <= MAX_BLOCK and offsets.numel() < block_count(n, block) + 1:
            raise ValueError("offsets needs nblocks+1 entries")
        st = lib().rcx_encode_blocks_device(self._h, coder, src.data_ptr(), n, block, dst.data_ptr(), dst.numel(),
                                            offsets.data_ptr(), self._stream_handle(stream))
        _check(st, "rcx_encode_blocks_device")

    def decode_blocks_device(self, comp, comp_size: int, offsets, n: int, block: int, out,
                             coder: int = CODER_ADAPTIVE, stream=None) -> None:
        nblocks = block_count(n, block)
        st = lib().rcx_decode_blocks_device(self._h, coder, comp.data_ptr(), comp_size, offsets.data_ptr(), nblocks, block,
                                            n, out.data_ptr(), self._stream_handle(stream))
        _check(st, "rcx_decode_blocks_device")

    def sync_status(self, stream=None, raise_on_error: bool = True):
        bad = C.c_uint64()
        st = lib().rcx_ctx_sync_status(self._h, self._stream_handle(stream), C.byref(bad))
        if st != OK and raise_on_error:
            raise RcxError(st, f"block {bad.value}")
        return st, bad.value

    # ---- host buffers ------------------------------------------------------
    def encode_blocks(self, data, block: int, coder: int = CODER_ADAPTIVE):
        """-> (payload uint8[total], offsets uint64[nblocks+1])"""
        src = _np_u8(data)
        n = len(src)
        nblocks = block_count(n, block)
        dst = np.empty(encode_bound(n, block, coder), dtype=np.uint8)
        offsets = np.zeros(nblocks + 1, dtype=np.uint64)
        size = C.c_uint64()
        st = lib().rcx_encode_blocks(self._h, coder, src.ctypes.data, n, block, dst.ctypes.data, len(dst), C.byref(size),
                                     offsets.ctypes.data)
        _check(st, "rcx_encode_blocks")
        return dst[: size.value], offsets

    def encode_blocks_into(self, src: np.ndarray, block: int, dst: np.ndarray, offsets: np.ndarray, coder: int = CODER_ADAPTIVE) -> int:
        """rcx_encode_blocks with the caller's buffers, as a C++ caller holds them (test/main.cpp:321-336 makes its
        streams before the clock starts): src uint8[n], dst uint8[>= encode_bound], offsets uint64[nblocks+1] -> size."""
        assert src.dtype == np.uint8 and dst.dtype == np.uint8 and offsets.dtype == np.uint64
        assert src.flags.c_contiguous and dst.flags.c_contiguous and offsets.flags.c_contiguous
        assert len(offsets) >= block_count(len(src), block) + 1
        size = C.c_uint64()
        st = lib().rcx_encode_blocks(self._h, coder, src.ctypes.data, len(src), block, dst.ctypes.data, len(dst), C.byref(size),
                                     offsets.ctypes.data)
        _check(st, "rcx_encode_blocks")
        return int(size.value)

    def decode_blocks_into(self, comp: np.ndarray, comp_size: int, offsets: np.ndarray, block: int, out: np.ndarray,
                           coder: int = CODER_ADAPTIVE) -> int:
        """rcx_decode_blocks with the caller's buffers -> decoded size."""
        assert comp.dtype == np.uint8 and out.dtype == np.uint8 and offsets.dtype == np.uint64
        assert comp.flags.c_contiguous and out.flags.c_contiguous and offsets.flags.c_contiguous
        size = C.c_uint64()
        st = lib().rcx_decode_blocks(self._h, coder, comp.ctypes.data, comp_size, offsets.ctypes.data, len(offsets) - 1, block,
                                     out.ctypes.data, len(out), C.byref(size))
        _check(st, "rcx_decode_blocks")
        return int(size.value)

    def decode_blocks(self, payload, offsets, block: int, capacity: int | None = None, coder: int = CODER_ADAPTIVE):
        comp = _np_u8(payload)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        nblocks = len(offs) - 1
        cap = nblocks * block if capacity is None else capacity
        out = np.empty(max(cap, 1), dtype=np.uint8)
        size = C.c_uint64()
        st = lib().rcx_decode_blocks(self._h, coder, comp.ctypes.data, len(comp), offs.ctypes.data, nblocks, block,
                                     out.ctypes.data, cap, C.byref(size))
        _check(st, "rcx_decode_blocks")
        return out[: size.value]

    # ---- block sort (blksort.h) ----------------------------------------------
    def bwt_encode(self, data) -> np.ndarray:
        """BlkSort::encode on a host buffer -> the encoded bytes."""
        src = _np_u8(data)
        dst = np.full(bwt_encode_bound(len(src)) + 64, 0xA5, dtype=np.uint8)
        size = C.c_uint64()
        _check(lib().rcx_bwt_encode(self._h, src.ctypes.data, len(src), dst.ctypes.data, len(dst) - 64, C.byref(size)), "rcx_bwt_encode")
        assert bool((dst[size.value:] == 0xA5).all()), "rcx_bwt_encode wrote past its size"
        return dst[: size.value]

    def bwt_decode(self, enc) -> np.ndarray:
        src = _np_u8(enc)
        dst = np.full(bwt_decoded_size(len(src)) + 64, 0xA5, dtype=np.uint8)
        size = C.c_uint64()
        _check(lib().rcx_bwt_decode(self._h, src.ctypes.data, len(src), dst.ctypes.data, len(dst) - 64, C.byref(size)), "rcx_bwt_decode")
        assert bool((dst[size.value:] == 0xA5).all()), "rcx_bwt_decode wrote past its size"
        return dst[: size.value]

    def bwt_into(self, forward: bool, src: np.ndarray, dst: np.ndarray) -> int:
        """rcx_bwt_encode / rcx_bwt_decode with the caller's buffers -> bytes written."""
        assert src.dtype == np.uint8 and dst.dtype == np.uint8 and src.flags.c_contiguous and dst.flags.c_contiguous
        size = C.c_uint64()
        fn = lib().rcx_bwt_encode if forward else lib().rcx_bwt_decode
        _check(fn(self._h, src.ctypes.data, len(src), dst.ctypes.data, len(dst), C.byref(size)), "rcx_bwt_encode" if forward else "rcx_bwt_decode")
        return int(size.value)

    def bwt_encode_device(self, src, dst, stream=None) -> None:
        """src, dst: uint8 cuda tensors (dst >= bwt_encode_bound(src.numel())); enqueues only."""
        _check(lib().rcx_bwt_encode_device(self._h, src.data_ptr(), src.numel(), dst.data_ptr(), dst.numel(), self._stream_handle(stream)),
               "rcx_bwt_encode_device")

    def bwt_decode_device(self, src, n: int, dst, stream=None) -> None:
        """src: uint8 cuda tensor holding n encoded bytes; dst >= bwt_decoded_size(n)."""
        _check(lib().rcx_bwt_decode_device(self._h, src.data_ptr(), n, dst.data_ptr(), dst.numel(), self._stream_handle(stream)),
               "rcx_bwt_decode_device")

    def bwt_last_ties(self) -> int:
        count = C.c_uint64(0)
        _check(lib().rcx_bwt_last_ties(self._h, C.byref(count)), "rcx_bwt_last_ties")
        return int(count.value)

    # ---- single streams (reference semantics, used by the C++ facade) ------
    def stream_encode(self, data, sink_capacity: int | None = None, coder: int = CODER_ADAPTIVE, dst_cap: int | None = None):
        """-> (status, request_size, stream bytes).  dst_cap: size of the buffer handed to the library (default: exactly
        what include/rcx.h says is needed, with canary bytes behind it that must survive)."""
        src = _np_u8(data)
        bound = block_bound(max(len(src), MIN_BLOCK), coder)
        cap = bound if sink_capacity is None else sink_capacity
        room = min(max(cap, 4) + 4, bound) if dst_cap is None else dst_cap
        dst = np.full(room + 64, 0xA5, dtype=np.uint8)
        size, req = C.c_uint64(), C.c_uint32()
        st = lib().rcx_stream_encode(self._h, coder, src.ctypes.data, len(src), dst.ctypes.data, room, cap, C.byref(size), C.byref(req))
        if not bool((dst[room:] == 0xA5).all()):
            raise AssertionError("rcx_stream_encode wrote past dst_cap")
        return st, req.value, bytes(dst[: min(size.value, room)]) if st in (OK, PENDING) else b""

    def stream_decode(self, comp, sink_capacity: int, coder: int = CODER_ADAPTIVE):
        src = _np_u8(comp)
        dst = np.zeros(max(sink_capacity, 16) + 64, dtype=np.uint8)
        size, req = C.c_uint64(), C.c_uint32()
        st = lib().rcx_stream_decode(self._h, coder, src.ctypes.data, len(src), dst.ctypes.data, sink_capacity, C.byref(size), C.byref(req))
        return st, req.value, bytes(dst[: size.value])

    # ---- the resumable decoder / encoder --------------------------------------
    def dstream(self) -> "DStream":
        return DStream(self)

    def estream(self, declared: int) -> "EStream":
        return EStream(self, declared)

    # ---- per-kernel device time -------------------------------------------
    def set_timing(self, enabled: bool) -> None:
        _check(lib().rcx_ctx_set_timing(self._h, int(enabled)), "rcx_ctx_set_timing")

    def last_redo(self, nblocks: int) -> int:
        """Blocks of the last encode/decode call that the one-lane kernels had to take over (0 on ordinary data)."""
        count = C.c_uint64(0)
        _check(lib().rcx_ctx_last_redo(self._h, C.c_uint64(nblocks), C.byref(count)), "rcx_ctx_last_redo")
        return int(count.value)

    def get_timing(self, reset: bool = True):
        ms = (C.c_double * T_COUNT)()
        launches = (C.c_uint64 * T_COUNT)()
        _check(lib().rcx_ctx_get_timing(self._h, ms, launches, int(reset)), "rcx_ctx_get_timing")
        names = ("encode", "scan", "scatter", "decode", "bwt_forward", "bwt_inverse")
        return {names[i]: {"ms": ms[i], "launches": int(launches[i])} for i in range(T_COUNT)}


class DStream:
    """One rcx_dstream: AdaptiveRangeDecoder<T>::decode fed piece by piece (cpprcoder.h:872-924)."""

    def __init__(self, ctx: Context):
        self._h = C.c_void_p()
        _check(lib().rcx_dstream_create(ctx._h, C.byref(self._h)), "rcx_dstream_create")

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().rcx_dstream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, piece, room: int):
        """-> (status, request_size, symbols decoded now)"""
        src = _np_u8(piece)
        dst = np.zeros(max(room, 1) + 16, dtype=np.uint8)
        got, req = C.c_uint64(), C.c_uint32()
        st = lib().rcx_dstream_decode(self._h, src.ctypes.data if len(src) else None, len(src), dst.ctypes.data, room, C.byref(got), C.byref(req))
        return st, req.value, bytes(dst[: got.value])


class EStream:
    """One rcx_estream: AdaptiveRangeEncoder<T>::encode fed piece by piece (cpprcoder.h:697-720)."""

    def __init__(self, ctx: Context, declared: int):
        self._h = C.c_void_p()
        _check(lib().rcx_estream_create(ctx._h, declared, C.byref(self._h)), "rcx_estream_create")

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().rcx_estream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, piece, sink_room: int | None = None):
        """-> (status, request_size, bytes for writeByte, bytes for write)"""
        src = _np_u8(piece)
        cap = 3 * len(src) + 64 + (1 << 16)
        while True:
            dst = np.full(cap + 16, 0xA5, dtype=np.uint8)
            got, tail, req = C.c_uint64(), C.c_uint32(), C.c_uint32()
            st = lib().rcx_estream_encode(self._h, src.ctypes.data if len(src) else None, len(src), dst.ctypes.data, cap,
                                          (1 << 64) - 1 if sink_room is None else sink_room, C.byref(got), C.byref(tail), C.byref(req))
            assert bool((dst[cap:] == 0xA5).all()), "rcx_estream_encode wrote past dst_cap"
            if st == E_CAPACITY:  # a pending run longer than the guess: once more with the size it asked for
                _check(lib().rcx_estream_rewind(self._h), "rcx_estream_rewind")
                cap = int(got.value) + 64
                continue
            body = int(got.value) - int(tail.value)
            return st, req.value, bytes(dst[:body]), bytes(dst[body: body + tail.value])

    def rewind(self) -> None:
        _check(lib().rcx_estream_rewind(self._h), "rcx_estream_rewind")
