"""A file container around the many-block coder ("multi-block container"; 8(f) row 3).

The reference has no multi-block format (one stream per file, test/main.cpp:304-364); this is the framing the
block engine needs to be usable on files.  Every block's stream inside is bit-exact what the reference's
AdaptiveRangeEncoder (or RangeEncoder, coder = 1; rANS::encode, coder = 2; rANS::encode_simd, coder = 3) emits
for that block, so a reader with only the reference can decode a container block by block.

Layout (little-endian):
    0   4  magic  b"RCXB"
    4   1  version (1)
    5   1  coder   (0 adaptive, 1 static, 2 rANS one state, 3 rANS eight states: include/rcx.h RCX_CODER_*)
    6   2  flags: bit 0 = the data went through the reference's block sort first (blksort.h: BlkSort::encode, 2 bytes
           more per whole 32 KiB, as test/main.cpp:961-970 does in front of zlib / zstd); other bits 0
    8   4  block size in bytes
    12  8  n, the original size
    20  8  nblocks = ceil(m / block), m = the bytes the coder saw: n, or with bit 0 n + 2 * (n // 32768)
    28  8 * (nblocks + 1)  offsets of the block streams in the payload (offsets[0] = 0, offsets[nblocks] = payload size)
    ..  payload: the block streams back to back

The header functions are plain Python; pack()/unpack() go through the HIP library (there is no CPU coder
here: without librcx.so and a GPU they raise).
"""
import struct

import numpy as np

MAGIC = b"RCXB"
VERSION = 1
FLAG_BLKSORT = 1
_FIXED = struct.Struct("<4sBBHIQQ")


def coded_size(n: int, flags: int) -> int:
    """What the entropy coder is handed for n original bytes (blksort.h:426-431 if block-sorted)."""
    return n + 2 * (n // 32768) if flags & FLAG_BLKSORT else n


class ContainerError(ValueError):
    pass


def header_bytes(coder: int, block: int, n: int, offsets, flags: int = 0) -> bytes:
    offsets = np.ascontiguousarray(offsets, dtype="<u8")
    nblocks = len(offsets) - 1
    if nblocks != (coded_size(n, flags) + block - 1) // block:
        raise ContainerError("offsets do not match n and the block size")
    return _FIXED.pack(MAGIC, VERSION, coder, flags, block, n, nblocks) + offsets.tobytes()


def parse(blob):
    """-> dict(coder, flags, block, n, nblocks, offsets uint64[nblocks+1], payload uint8 view)"""
    buf = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    if len(buf) < _FIXED.size:
        raise ContainerError("shorter than a header")
    magic, version, coder, flags, block, n, nblocks = _FIXED.unpack(bytes(buf[: _FIXED.size]))
    if magic != MAGIC:
        raise ContainerError("not an RCXB container")
    if version != VERSION or coder not in (0, 1, 2, 3) or flags & ~FLAG_BLKSORT:
        raise ContainerError("unsupported container version, coder or flags")
    if block < 16 or block > (1 << 24) - 256 or nblocks != (coded_size(n, flags) + block - 1) // block:
        raise ContainerError("inconsistent header")
    end = _FIXED.size + 8 * (nblocks + 1)
    if len(buf) < end:
        raise ContainerError("truncated offset table")
    offsets = np.frombuffer(bytes(buf[_FIXED.size:end]), dtype="<u8").astype(np.uint64)
    if offsets[0] != 0 or np.any(np.diff(offsets.astype(np.int64)) < 0) or end + int(offsets[-1]) != len(buf):
        raise ContainerError("offset table does not match the payload")
    return {"coder": coder, "flags": flags, "block": block, "n": n, "nblocks": nblocks, "offsets": offsets, "payload": buf[end:]}


def pack(data, block: int = 65536, coder: int = 0, ctx=None, blksort: bool = False) -> bytes:
    from . import rcx
    own = ctx is None
    ctx = ctx or rcx.Context(0)
    try:
        src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        flags = FLAG_BLKSORT if blksort else 0
        if len(src) == 0:
            return header_bytes(coder, block, 0, np.zeros(1, np.uint64), flags)
        payload, offsets = ctx.encode_blocks(ctx.bwt_encode(src) if blksort else src, block, coder=coder)
        return header_bytes(coder, block, len(src), offsets, flags) + payload.tobytes()
    finally:
        if own:
            ctx.close()


def unpack(blob, ctx=None) -> bytes:
    from . import rcx
    c = parse(blob)
    if c["n"] == 0:
        return b""
    own = ctx is None
    ctx = ctx or rcx.Context(0)
    try:
        m = coded_size(c["n"], c["flags"])
        out = ctx.decode_blocks(c["payload"], c["offsets"], c["block"], capacity=m, coder=c["coder"])
        if len(out) != m:
            raise ContainerError("decoded size differs from the header")
        if c["flags"] & FLAG_BLKSORT:
            out = ctx.bwt_decode(out)
        return out.tobytes()
    finally:
        if own:
            ctx.close()
