"""python -m cpprcoder_amd c|d|t ...  -- compress / decompress / test files with the MI355X block coder.

    python -m cpprcoder_amd c [-b BLOCK] [--blksort] [--static | --coder adaptive|static|rans|rans8] IN OUT
                                                               IN -> RCXB container (cpprcoder_amd/container.py);
                                                               --blksort: the reference's block sort (blksort.h) first
    python -m cpprcoder_amd d IN OUT                           container -> original bytes
    python -m cpprcoder_amd t FILE...                          the reference harness's row per file
                                                               (|file|ratio|encode|decode|, test/main.cpp:346-356):
                                                               pack, unpack, compare, times incl. PCIe copies
"""
import argparse
import sys
import time


CODERS = ("adaptive", "static", "rans", "rans8")  # include/rcx.h: RCX_CODER_*


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m cpprcoder_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("c")
    c.add_argument("-b", "--block", type=int, default=65536)
    c.add_argument("--static", action="store_true")
    c.add_argument("--coder", choices=CODERS, default=None)
    c.add_argument("--blksort", action="store_true")
    c.add_argument("src")
    c.add_argument("dst")
    d = sub.add_parser("d")
    d.add_argument("src")
    d.add_argument("dst")
    t = sub.add_parser("t")
    t.add_argument("-b", "--block", type=int, default=65536)
    t.add_argument("--static", action="store_true")
    t.add_argument("--coder", choices=CODERS, default=None)
    t.add_argument("--blksort", action="store_true")
    t.add_argument("files", nargs="+")
    a = ap.parse_args(argv)
    from . import container, rcx
    coder = CODERS.index(a.coder) if getattr(a, "coder", None) else (1 if getattr(a, "static", False) else 0)
    ctx = rcx.Context(0)
    try:
        if a.cmd == "c":
            data = open(a.src, "rb").read()
            blob = container.pack(data, a.block, coder, ctx, blksort=a.blksort)
            open(a.dst, "wb").write(blob)
            print(f"{a.src}: {len(data)} -> {len(blob)} bytes ({len(blob) / max(len(data), 1):.6f})")
        elif a.cmd == "d":
            out = container.unpack(open(a.src, "rb").read(), ctx)
            open(a.dst, "wb").write(out)
            print(f"{a.src}: {len(out)} bytes")
        else:
            print("|file|ratio|encode (microseconds)|decode (microseconds)|")
            print("|:---|:---|:---|:---|")
            bad = 0
            for path in a.files:
                data = open(path, "rb").read()
                t0 = time.perf_counter()
                blob = container.pack(data, a.block, coder, ctx, blksort=a.blksort)
                t1 = time.perf_counter()
                back = container.unpack(blob, ctx)
                t2 = time.perf_counter()
                ok = back == data
                bad += not ok
                print(f"|{path}|{len(blob) / max(len(data), 1):.6f}|{(t1 - t0) * 1e6:.0f}|{(t2 - t1) * 1e6:.0f}|" + ("" if ok else " MISMATCH"))
            return 1 if bad else 0
    finally:
        ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
