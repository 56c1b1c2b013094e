"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/rcx.h declares.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from cpprcoder_amd import build, rcx
    build.build()
    return rcx.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rcx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rcx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    from cpprcoder_amd import rcx
    names = declared_symbols()
    assert names and set(names) == set(rcx.EXPORTS), (names, rcx.EXPORTS)
    for name in names:
        assert getattr(lib, name) is not None


def test_geometry_helpers(lib):
    from cpprcoder_amd import rcx
    assert lib.rcx_version() == 300
    assert rcx.block_count(0, 65536) == 0 and rcx.block_count(1, 65536) == 1 and rcx.block_count(1 << 30, 65536) == 16384
    assert rcx.block_count(65537, 65536) == 2
    for block in (16, 4096, 65536, 1 << 20, 2 << 20, rcx.MAX_BLOCK):
        bound = rcx.block_bound(block)
        assert bound % 16 == 0 and bound >= block + 521 + 9  # room for the static coder's table header too
        assert rcx.encode_bound(10 * block + 1, block) >= 11 * bound
    assert rcx.status_string(0) == "success" and "corrupt" in rcx.status_string(rcx.E_CORRUPT)


def test_no_gpu_means_loud_failure(lib):
    import torch
    from cpprcoder_amd import rcx
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rcx.RcxError):  # no CPU fallback
        rcx.Context(0)


def test_product_does_not_link_the_oracle():
    # the product path must never route through oracle/: neither the sources nor the binary mention it
    from cpprcoder_amd import rcx
    pkg = os.path.join(ROOT, "cpprcoder_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "rc_oracle" not in text and "oracle_lib" not in text, f
    blob = open(rcx.LIB_PATH, "rb").read()
    assert b"rco_" not in blob and b"liboracle" not in blob
