#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --stats` CSV -> the text summary committed under profiles/ (per kernel instantiation, plus one row
per kernel family with all its template instantiations merged: what bench.py's `roofline.avg_launch_ms` is compared with).
Usage: python tools/stats_summary.py <st_kernel_stats.csv> <n_steps in trace> "<header comment>" > profiles/rNN_kernel_stats.txt"""
import csv
import re
import sys
from collections import OrderedDict


def main():
    path, steps, note = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    print(f"# {note}")
    print(f"# total kernel time {tot:.1f} ms over {steps} steps = {tot / steps:.2f} ms of kernel time per step")
    fam = OrderedDict()
    for r in rows:
        m = re.search(r"cdet::([A-Za-z0-9_]+)", r["Name"])
        k = m.group(1) if m else "(other) " + r["Name"][:40]
        f = fam.setdefault(k, [0, 0.0])
        f[0] += int(r["Calls"])
        f[1] += float(r["TotalDurationNs"]) / 1e6
    print(f"{'kernel family (all template instantiations)':60s} {'calls':>7s} {'total ms':>10s} {'ms/step':>8s} {'avg us':>9s} {'%':>6s}")
    for k, (n, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{k:60s} {n:7d} {ms:10.2f} {ms / steps:8.3f} {ms / n * 1e3:9.1f} {100 * ms / tot:6.2f}")
    print()
    print(f"{'kernel':100s} {'calls':>7s} {'total ms':>10s} {'ms/step':>8s} {'avg us':>9s} {'%':>6s}")
    for r in rows[:40]:
        ms = float(r["TotalDurationNs"]) / 1e6
        print(f"{r['Name'][:100]:100s} {int(r['Calls']):7d} {ms:10.2f} {ms / steps:8.3f} {float(r['AverageNs']) / 1e3:9.1f} {100 * ms / tot:6.2f}")


if __name__ == "__main__":
    main()
