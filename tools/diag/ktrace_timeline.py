"""Timeline of the last host-buffer encode and decode in a rocprofv3 --kernel-trace --memory-copy-trace CSV directory."""
import csv
import glob
import sys

d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/*_kernel_trace.csv")[0])))
cop = list(csv.DictReader(open(glob.glob(d + "/*_memory_copy_trace.csv")[0])))
for kernel in ("rcx_enc_mc5_k", "rcx_dec_quad_k"):
    mine = [r for r in rows if kernel in r["Kernel_Name"]]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    last = mine[-n:]
    t0 = int(last[0]["Start_Timestamp"])
    lo, hi = t0 - 6_000_000, int(last[-1]["End_Timestamp"]) + 8_000_000
    ev = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= s <= hi:
            nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
            ev.append((s, e, "K q%s s%s %s grid %s" % (r["Queue_Id"], r["Stream_Id"], nm, r["Grid_Size_X"])))
    for r in cop:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= s <= hi:
            ev.append((s, e, "C %s s%s" % (r["Direction"], r["Stream_Id"])))
    ev.sort()
    print("=====", kernel)
    prev = None
    for s, e, dsc in ev:
        if "scan" in dsc or "scatter" in dsc or "adaptive" in dsc:
            continue
        key = dsc
        if prev and prev[2] == key and (dsc.startswith("C ") or "copyBuffer" in dsc):  # merge runs of equal copies
            prev[1] = e
            prev[3] += 1
            continue
        if prev:
            print(f"{(prev[0]-t0)/1e6:8.2f} -> {(prev[1]-t0)/1e6:8.2f}  x{prev[3]:<3d} {prev[2]}")
        prev = [s, e, key, 1]
    if prev:
        print(f"{(prev[0]-t0)/1e6:8.2f} -> {(prev[1]-t0)/1e6:8.2f}  x{prev[3]:<3d} {prev[2]}")
