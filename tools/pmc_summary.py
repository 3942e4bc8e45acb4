#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs into per-kernel HBM traffic per launch.

    python tools/pmc_summary.py FETCH_CSV WRITE_CSV > profiles/rN_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B... (rocprofv3: kilobytes); per MI355X_MICROARCH.md §HBM,
on gfx950 FETCH_SIZE reads exactly HALF of the bytes of a wide coalesced streaming read, so it is doubled here;
WRITE_SIZE is exact for 16-byte-per-lane stores. The two counters need separate passes (TCC slots)."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(conv_ring2_kernel|conv_ring_kernel|conv_igemm_kernel|conv1x1_stream_kernel)<([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(2).split(",")]
        dt = {"unsigned short": "bf16", "_Float16": "f16", "float": "f32"}.get(args[0], args[0])
        if m.group(1) == "conv_ring_kernel":
            two = len(args) > 3 and args[3] in ("true", "1")      # round 3: the two-source form is a template instance of its own
            return "conv_ring_kernel<%dx%d%s>[%s]" % (32 * int(args[2]), 64 * int(args[1]), ",2src" if two else "", dt)
        if m.group(1) == "conv_ring2_kernel":
            return "conv_ring2_kernel<256x128>[%s]" % dt
        if m.group(1) == "conv1x1_stream_kernel":
            return "conv1x1_stream_kernel[%s]" % dt
        return "conv_igemm_kernel<%s>[%s]" % (args[1], dt)
    m = re.search(r"conv3x3_wsr_kernel<([^>]*)>", name)
    if m:   # <storage type, channels, rows per step, relu>: the class names engine.py / bench.py use
        args = [a.strip() for a in m.group(1).split(",")]
        dt = {"unsigned short": "bf16", "_Float16": "f16"}.get(args[0], args[0])
        post = ",post%s" % args[4] if len(args) > 4 and args[4] not in ("0",) else ""
        return "conv3x3_wsr_kernel<%s,%s%s>[%s]" % (args[1], "relu" if args[3] in ("true", "1") else "linear", post, dt)
    m = re.search(r"(conv3x3_rows2?_kernel)<([^>]*)>", name)
    if m:   # <storage type, input channels, ...>: one line per channel count, as engine.py / bench.py name them
        args = [a.strip() for a in m.group(2).split(",")]
        return "%s<%s>" % (m.group(1), args[1]) if len(args) > 1 else m.group(1)
    m = re.search(r"bottleneck_tail64_kernel<([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(1).split(",")]
        dt = {"unsigned short": "bf16", "_Float16": "f16"}.get(args[0], args[0])
        return "bottleneck_tail64_kernel<%s>[%s]" % ("next" if args[1] == "true" else "last", dt)
    m = re.search(r"::(\w+)(<|\()", name)
    if m:
        return m.group(1)
    m = re.search(r"(\w+)(<|\()", name)
    return m.group(1) if m else name[:60]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])
        w = write.get(k, [])
        fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)   # gfx950 correction x2
        wb = 1024.0 * sum(w) / max(len(w), 1)
        out[k] = {"launches_profiled": max(len(f), len(w)), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                  "hbm_bytes_per_launch": round(fb + wb)}
    json.dump({"note": "FETCH_SIZE x2 (gfx950 under-reports wide streaming reads by 1/2), WRITE_SIZE as is; units: bytes per launch, "
                       "averaged over the launches of each kernel in `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graphs --streams 1`",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
