#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (CSV) into per-kernel HBM traffic per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of wide coalesced streaming reads -> doubled here. WRITE_SIZE is uncalibrated (taken at face value).
"""
import collections
import csv
import json
import re
import sys


def agg(d, counter):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{d}/pmc_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def short(name):
    m = re.search(r"cdet::([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    fd, wd, dst = sys.argv[1:4]
    f, w = agg(fd, "FETCH_SIZE"), agg(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if "cdet::" not in k:
            continue
        fv, wv = f.get(k, []), w.get(k, [])
        res[short(k)] = dict(launches=len(fv) or len(wv),
                             fetch_bytes_per_launch=round(2 * 1024 * sum(fv) / max(len(fv), 1)),
                             write_bytes_per_launch=round(1024 * sum(wv) / max(len(wv), 1)),
                             fetch_bytes_total=round(2 * 1024 * sum(fv)), write_bytes_total=round(1024 * sum(wv)))
    json.dump(dict(note="FETCH_SIZE x2 (gfx950 correction) x1024, WRITE_SIZE x1024; per launch = mean over all launches of the kernel",
                   kernels=res), open(dst, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["fetch_bytes_total"] + kv[1]["write_bytes_total"]))[:12]:
        print(f"{k:60s} n={v['launches']:5d} fetch/launch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write/launch {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
