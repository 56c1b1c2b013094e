#!/usr/bin/env python3
"""Turn rocprofv3 CSV output under gpurun_out/ into the small tracked summaries in profiles/.

    python profiles/summarize.py <tag> <stats_dir> [--side] [--config workload=uniform,bytes=1073741824,block=65536] [--pmc name=dir ...]

--side: a profile of something other than bench.py's default line (e.g. the block sort): profiles/pmc_traffic.json, which
bench.py replays into roofline.traffic, is left alone.

<stats_dir> is a `rocprofv3 --kernel-trace --stats --output-format csv -d <dir>` directory;
each --pmc dir is a separate `rocprofv3 --pmc ... --kernel-trace` pass (FETCH_SIZE and WRITE_SIZE
must come from different passes: they do not fit the TCC counter slots together).
"""
import collections
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name: str) -> str:
    name = name.replace("void ", "")
    base = name.split("(")[0]
    if base.startswith("rcx_"):
        return base.split("<")[0]  # rcx_dec_quad_k<4> -> rcx_dec_quad_k
    return base[:48] + "...(torch)"


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    side = bool(rest) and rest[0] == "--side"
    if side:
        rest = rest[1:]
    config = {"workload": "uniform", "bytes": 1 << 30, "block": 65536}  # what bench.py runs by default
    if rest and rest[0] == "--config":
        for kv in rest[1].split(","):
            k, v = kv.split("=", 1)
            config[k] = int(v) if v.isdigit() else v
        rest = rest[2:]
    config["recorded"] = tag
    pmc_dirs = dict(a.split("=", 1) for a in rest[1:]) if rest and rest[0] == "--pmc" else {}
    newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
    stats = newest(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(HERE, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            r = dict(r)
            r["Name"] = short(r["Name"])
            w.writerow(r)
    counters = collections.defaultdict(dict)
    for label, d in pmc_dirs.items():
        path = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            if "rcx_" not in r["Kernel_Name"]:
                continue
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            for name, v in c.items():
                counters[k][name] = sum(v) / len(v)
    out = {"_config": config,
           "_how": ("per-launch averages from separate rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline` "
                    "on the workload in _config. FETCH_SIZE / WRITE_SIZE are KiB; per MI355X_MICROARCH.md the gfx950 FETCH_SIZE "
                    "is doubled (128-B requests are tallied at 64 B), WRITE_SIZE is taken as is. That correction is calibrated for "
                    "16-B-per-lane coalesced streams; these kernels read and write lane-strided pieces, so the absolute is approximate.")}
    for k, c in counters.items():
        e = {"counters_per_launch": c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["fetch_bytes_corrected"] = int(2 * c["FETCH_SIZE"] * 1024)
            e["write_bytes"] = int(c["WRITE_SIZE"] * 1024)
            e["hbm_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
        out[k] = e
    if counters:
        with open(os.path.join(HERE, f"{tag}_pmc.json"), "w") as f:
            json.dump(out, f, indent=1)
        # bench.py reads roofline.traffic from here
        if not side:
            with open(os.path.join(HERE, "pmc_traffic.json"), "w") as f:
                json.dump(out, f, indent=1)
    for r in rows[:6]:
        print(short(r["Name"]), r["Calls"], r["AverageNs"])
    for k, e in out.items():
        if not k.startswith("_") and "hbm_bytes_per_launch" in e:
            print(k, "HBM GiB/launch %.3f" % (e["hbm_bytes_per_launch"] / 2 ** 30))


if __name__ == "__main__":
    main()
