#!/usr/bin/env python3
"""Which launches surround the runtime's own copy kernels (__amd_rocclr_copyBuffer) in a rocprofv3 --kernel-trace CSV:
prints (previous kernel, next kernel) pairs on the same queue with counts. Usage: copy_neighbors.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
short = lambda n: n.split("(")[0].replace("void cdet::", "")[:60]  # noqa: E731
cnt = collections.Counter()
sizes = collections.Counter()
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]:
            p = short(rs[i - 1]["Kernel_Name"]) if i else "-"
            n = short(rs[i + 1]["Kernel_Name"]) if i + 1 < len(rs) else "-"
            cnt[(short(r["Kernel_Name"]), p, n)] += 1
            sizes[(r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")))] += 1
for k, v in cnt.most_common(40):
    print(v, k)
print(sizes.most_common(10))
