"""The RCXB container: header logic on the CPU, pack/unpack + the CLI on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cpprcoder_amd import container

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_round_trip_and_rejections():
    offs = np.array([0, 100, 250, 251], np.uint64)
    h = container.header_bytes(0, 4096, 3 * 4096 - 7, offs)
    blob = h + bytes(251)
    c = container.parse(blob)
    assert (c["coder"], c["block"], c["n"], c["nblocks"]) == (0, 4096, 3 * 4096 - 7, 3)
    assert np.array_equal(c["offsets"], offs) and len(c["payload"]) == 251
    with pytest.raises(container.ContainerError):
        container.parse(b"nope" + blob[4:])
    with pytest.raises(container.ContainerError):
        container.parse(blob[:-1])                      # payload shorter than the table says
    with pytest.raises(container.ContainerError):
        container.parse(blob[:30])                      # truncated table
    with pytest.raises(container.ContainerError):
        container.header_bytes(0, 4096, 5 * 4096, offs)  # wrong block count
    bad = bytearray(blob)
    bad[5] = 7                                          # unknown coder (0..3 are include/rcx.h's RCX_CODER_*)
    with pytest.raises(container.ContainerError):
        container.parse(bytes(bad))
    empty = container.parse(container.header_bytes(1, 65536, 0, np.zeros(1, np.uint64)))
    assert empty["nblocks"] == 0 and len(empty["payload"]) == 0
    # block-sorted first (flags bit 0): the coder saw 2 bytes more per whole 32 KiB (blksort.h:426-431)
    n = 3 * 32768 - 2
    assert container.coded_size(n, container.FLAG_BLKSORT) == n + 4
    offs = np.array([0, 10, 20, 30, 40], np.uint64)                   # 98306 bytes in blocks of 32768: 4 blocks, not 3
    c = container.parse(container.header_bytes(0, 32768, n, offs, container.FLAG_BLKSORT) + bytes(40))
    assert c["flags"] == container.FLAG_BLKSORT and c["nblocks"] == 4
    with pytest.raises(container.ContainerError):
        container.header_bytes(0, 32768, n, offs)                     # without the flag the same table is wrong
    bad = bytearray(container.header_bytes(0, 4096, 100, np.array([0, 5], np.uint64)) + bytes(5))
    bad[6] = 2                                                         # an unknown flag
    with pytest.raises(container.ContainerError):
        container.parse(bytes(bad))


@pytest.mark.gpu
def test_pack_unpack_and_blocks_are_the_references(oracle):
    from cpprcoder_amd import rcx, workloads
    ctx = rcx.Context(0)
    try:
        for coder in (0, 1, 2, 3):
            for n, block in ((0, 65536), (1, 4096), (300_001, 4096), (2_000_000, 65536)):
                data = workloads.zipf(n, 3 + n % 7) if n else np.zeros(0, np.uint8)
                blob = container.pack(data, block, coder, ctx)
                c = container.parse(blob)
                assert c["n"] == n and c["coder"] == coder
                if n:
                    slots, sizes = oracle.encode_blocks(data, block, coder=coder, threads=4)
                    assert np.array_equal(np.diff(c["offsets"].astype(np.int64)), sizes.astype(np.int64))
                    for b in (0, len(sizes) // 2, len(sizes) - 1):
                        got = c["payload"][int(c["offsets"][b]): int(c["offsets"][b + 1])]
                        assert np.array_equal(got, slots[b, : int(sizes[b])]), (coder, n, block, b)
                assert container.unpack(blob, ctx) == data.tobytes()
        # block sort in front of the coder: the coder's input is BlkSort::encode's output
        data = workloads.canterbury_concat()[:1_000_000]
        plain = container.pack(data, 65536, 0, ctx)
        blob = container.pack(data, 65536, 0, ctx, blksort=True)
        c = container.parse(blob)
        assert c["flags"] == container.FLAG_BLKSORT and c["n"] == len(data)
        sorted_first = oracle.bwt_encode(data, threads=8)
        slots, sizes = oracle.encode_blocks(sorted_first, 65536, threads=4)
        assert np.array_equal(np.diff(c["offsets"].astype(np.int64)), sizes.astype(np.int64))
        assert container.unpack(blob, ctx) == data.tobytes()
        # The transform permutes each block's bytes: an order-0 model sees the same counts, so the size barely moves
        # (the reference pairs its block sort with zlib / zstd, test/main.cpp:961-970, and its own move-to-front stage
        # is compiled out, blksort.h:54).
        assert abs(len(blob) - len(plain)) < 0.01 * len(plain)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_cli_round_trip(tmp_path):
    from cpprcoder_amd import workloads
    src = tmp_path / "in.bin"
    src.write_bytes(workloads.canterbury_concat()[:700_000].tobytes())
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.run([sys.executable, "-m", "cpprcoder_amd", *a], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    r = run("c", "-b", "16384", str(src), str(tmp_path / "out.rcxb"))
    assert r.returncode == 0, r.stderr
    r = run("d", str(tmp_path / "out.rcxb"), str(tmp_path / "back.bin"))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "back.bin").read_bytes() == src.read_bytes()
    r = run("t", "--static", str(src))
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout + r.stderr
    r = run("t", "--coder", "rans8", "-b", "16384", str(src))
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout + r.stderr
    assert (tmp_path / "out.rcxb").stat().st_size < 0.6 * src.stat().st_size
    r = run("c", "--blksort", str(src), str(tmp_path / "sorted.rcxb"))
    assert r.returncode == 0, r.stderr
    r = run("d", str(tmp_path / "sorted.rcxb"), str(tmp_path / "back2.bin"))
    assert r.returncode == 0 and (tmp_path / "back2.bin").read_bytes() == src.read_bytes()
