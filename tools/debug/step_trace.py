"""Concurrency accounting of one training iteration from a rocprofv3 kernel trace of bench.py (two task streams):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st -o st -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-infer --no-breakdown
  python3 tools/debug/step_trace.py gpurun_out/st > profiles/rNN_step_timeline.txt
Takes the second-to-last iteration (delimited by sgd_ema_kernel), prints how long 0 / 1 / 2 / 3+ kernels were in flight, which kernel families ran
ALONE (time with exactly one kernel in flight, by family), and the idle gaps by the kernel that follows them."""
import csv
import glob
import re
import sys

fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))


def fam(n):
    m = re.search(r"cdet::([A-Za-z0-9_]+)", n)
    return m.group(1) if m else n[:40]


ends = [i for i, r in enumerate(rows) if "sgd_ema_kernel" in r["Kernel_Name"]]
lo, hi = ends[-3] + 1, ends[-2] + 1
st = rows[lo:hi]
t0 = int(st[0]["Start_Timestamp"])
ev = []
for k, r in enumerate(st):
    ev.append((int(r["Start_Timestamp"]) - t0, 1, k))
    ev.append((int(r["End_Timestamp"]) - t0, -1, k))
ev.sort()
hist, alone, gap_before = {}, {}, {}
live = set()
last = 0
for t, d, k in ev:
    n = len(live)
    dt = t - last
    hist[min(n, 3)] = hist.get(min(n, 3), 0) + dt
    if n == 1:
        f = fam(st[next(iter(live))]["Kernel_Name"])
        alone[f] = alone.get(f, 0) + dt
    if n == 0 and d == 1 and dt > 0:
        f = fam(st[k]["Kernel_Name"])
        g = gap_before.setdefault(f, [0, 0])
        g[0] += dt
        g[1] += 1
    last = t
    if d == 1:
        live.add(k)
    else:
        live.discard(k)
span = ev[-1][0]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st)
print(f"# one iteration: {len(st)} dispatches, span {span / 1e6:.2f} ms, sum of kernel durations {tot / 1e6:.2f} ms")
for n in sorted(hist):
    print(f"# {n}{'+' if n == 3 else ''} kernels in flight: {hist[n] / 1e6:7.2f} ms")
print("# time with exactly ONE kernel in flight, by family (ms):")
for f, v in sorted(alone.items(), key=lambda kv: -kv[1])[:16]:
    print(f"#   {f:34s} {v / 1e6:6.2f}")
print("# idle gaps (no kernel in flight) by the kernel that ends them (ms, count):")
for f, (v, c) in sorted(gap_before.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"#   {f:34s} {v / 1e6:6.2f} {c:5d}")

# per-queue busy time, and (argv[2] = "dump <from_ms> <to_ms>") the dispatches of a window
qb = {}
for r in st:
    q = r.get("Queue_Id", "?")
    qb[q] = qb.get(q, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# busy time per hardware queue (ms): " + ", ".join(f"q{q}: {v / 1e6:.2f}" for q, v in sorted(qb.items())))
if len(sys.argv) > 4 and sys.argv[2] == "dump":
    a, b = float(sys.argv[3]) * 1e6, float(sys.argv[4]) * 1e6
    for r in st:
        s_, e_ = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if e_ >= a and s_ <= b:
            wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) if "Grid_Size_X" in r else 0
            print(f"{s_ / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f} q{r.get('Queue_Id', '?'):>3s} {wg:6d}  {fam(r['Kernel_Name'])}")
