"""MI355X-native many-block adaptive range coder behind cpprcoder.h's byte-stream format.

Layout:
    csrc/        hand-written HIP for gfx950 + the C ABI (include/rcx.h) -> librcx.so
    rcx.py       ctypes binding of the C ABI (host plumbing for tests / bench.py)
    workloads.py seeded synthetic inputs of BASELINE.json's configs
    build.py     in-tree hipcc build
The C++ host facade with the reference's class names lives in include/cpprcoder_amd/cpprcoder.h.
"""
__version__ = "0.1.0"
