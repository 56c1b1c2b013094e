"""Build the native library in-tree: hipcc --offload-arch=gfx950 -> cpprcoder_amd/librcx.so.

hipcc cross-compiles gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librcx.so")
LIB_VARIANTS = os.path.join(HERE, "librcx_variants.so")  # diagnostic: + the superseded kernels of csrc/variants/ (RCX_LIBRARY=...)
SOURCES = ["rcx_api.hip", "rcx_comm.hip"]
HEADERS = ["rcx_lane.hpp", "rcx_divtab.hpp", "rcx_kernels.hpp", "rcx_oct.hpp", "rcx_static.hpp", "rcx_rans.hpp", "rcx_bwt.hpp", "rcx_bwt_tie.hpp", os.path.join("variants", "rcx_variants.hpp"), "rcx_comm.hip", os.path.join("..", "..", "include", "rcx.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, variants: bool = False) -> str:
    """librcx.so, the product; variants=True: librcx_variants.so, the same + the superseded kernels (csrc/variants/), which the
    product does not carry -- tests/test_gpu_parity.py checks them through it, tools compare against them."""
    lib = LIB_VARIANTS if variants else LIB
    if not force and not stale(lib):
        return lib
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + (["-DRCX_WITH_VARIANTS"] if variants else []) + [
           "-o", lib] + [os.path.join(CSRC, s) for s in SOURCES] + ["-L/opt/rocm/lib", "-lrccl"]  # rcx_comm.hip: RCCL
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(" ".join(cmd))
        print(proc.stdout, proc.stderr)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stderr[-4000:])
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, variants="--variants" in sys.argv))
