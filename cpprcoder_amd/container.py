"""A file container around the many-block coder ("multi-block container"; 8(f) row 3).

The reference has no multi-block format (one stream per file, test/main.cpp:304-364); this is the framing the
block engine needs to be usable on files.  Every block's stream inside is bit-exact what the reference's
AdaptiveRangeEncoder (or RangeEncoder, coder = 1; rANS::encode, coder = 2; rANS::encode_simd, coder = 3) emits
for that block, so a reader with only the reference can decode a container block by block.

Layout (little-endian):
    0   4  magic  b"RCXB"
    4   1  version (1)
    5   1  coder   (0 adaptive, 1 static, 2 rANS one state, 3 rANS eight states: include/rcx.h RCX_CODER_*)
    6   2  reserved (0)
    8   4  block size in bytes
    12  8  n, the original size
    20  8  nblocks = ceil(n / block)
    28  8 * (nblocks + 1)  offsets of the block streams in the payload (offsets[0] = 0, offsets[nblocks] = payload size)
    ..  payload: the block streams back to back

The header functions are plain Python; pack()/unpack() go through the HIP library (there is no CPU coder
here: without librcx.so and a GPU they raise).
"""
import struct

import numpy as np

MAGIC = b"RCXB"
VERSION = 1
_FIXED = struct.Struct("<4sBBHIQQ")


class ContainerError(ValueError):
    pass


def header_bytes(coder: int, block: int, n: int, offsets) -> bytes:
    offsets = np.ascontiguousarray(offsets, dtype="<u8")
    nblocks = len(offsets) - 1
    if nblocks != (n + block - 1) // block:
        raise ContainerError("offsets do not match n and the block size")
    return _FIXED.pack(MAGIC, VERSION, coder, 0, block, n, nblocks) + offsets.tobytes()


def parse(blob):
    """-> dict(coder, block, n, nblocks, offsets uint64[nblocks+1], payload uint8 view)"""
    buf = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    if len(buf) < _FIXED.size:
        raise ContainerError("shorter than a header")
    magic, version, coder, reserved, block, n, nblocks = _FIXED.unpack(bytes(buf[: _FIXED.size]))
    if magic != MAGIC:
        raise ContainerError("not an RCXB container")
    if version != VERSION or coder not in (0, 1, 2, 3) or reserved != 0:
        raise ContainerError("unsupported container version or coder")
    if block < 16 or block > (1 << 24) - 256 or nblocks != (n + block - 1) // block:
        raise ContainerError("inconsistent header")
    end = _FIXED.size + 8 * (nblocks + 1)
    if len(buf) < end:
        raise ContainerError("truncated offset table")
    offsets = np.frombuffer(bytes(buf[_FIXED.size:end]), dtype="<u8").astype(np.uint64)
    if offsets[0] != 0 or np.any(np.diff(offsets.astype(np.int64)) < 0) or end + int(offsets[-1]) != len(buf):
        raise ContainerError("offset table does not match the payload")
    return {"coder": coder, "block": block, "n": n, "nblocks": nblocks, "offsets": offsets, "payload": buf[end:]}


def pack(data, block: int = 65536, coder: int = 0, ctx=None) -> bytes:
    from . import rcx
    own = ctx is None
    ctx = ctx or rcx.Context(0)
    try:
        src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        if len(src) == 0:
            return header_bytes(coder, block, 0, np.zeros(1, np.uint64))
        payload, offsets = ctx.encode_blocks(src, block, coder=coder)
        return header_bytes(coder, block, len(src), offsets) + payload.tobytes()
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
        out = ctx.decode_blocks(c["payload"], c["offsets"], c["block"], capacity=c["n"], coder=c["coder"])
        if len(out) != c["n"]:
            raise ContainerError("decoded size differs from the header")
        return out.tobytes()
    finally:
        if own:
            ctx.close()
