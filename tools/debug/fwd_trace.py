"""Timeline of one north-star forward (YOLOv8x 2-task all heads, eval, bf16, batch 32 @640) from a rocprofv3 kernel trace.
  run   : rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft -o ft -- python3 tools/debug/fwd_trace.py run
  parse : python3 tools/debug/fwd_trace.py parse gpurun_out/ft > profiles/rNN_fwd_timeline.txt
The parse step takes the LAST forward of the trace (delimited by the first-row kernel of the backbone) and prints every dispatch with its
start offset, duration, queue and grid, then the busy / idle / overlap accounting."""
import csv, glob, os, re, sys

if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    import bench

    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model("v8x_2task.yaml", dev)
    x = torch.rand(32, 3, 640, 640).bfloat16().to(dev)
    model.eval().bfloat16()
    with torch.no_grad():
        for _ in range(40):
            model(x)
        torch.cuda.synchronize()
    sys.exit(0)

fs = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"cdet::([A-Za-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:50]


first = [i for i, r in enumerate(rows) if "stem" in r["Kernel_Name"]]
lo, hi = first[-2], first[-1]  # the second-to-last forward (complete)
fw = rows[lo:hi]
t0 = int(fw[0]["Start_Timestamp"])
ev = []
print(f"# {len(fw)} dispatches; columns: start us, duration us, queue, workgroups, kernel")
for r in fw:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) if "Grid_Size_X" in r else 0
    q = r.get("Queue_Id", "?")
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q:>3s} {wg:6d}  {short(r['Kernel_Name'])}")
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = idle = 0
hist = {}
cur, last = 0, ev[0][0]
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    last = t
    cur += d
span = ev[-1][0] - ev[0][0]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw)
nxt = int(rows[hi]["Start_Timestamp"]) - t0
print(f"# span {span / 1e6:.3f} ms (next forward starts at {nxt / 1e6:.3f} ms), sum of kernel durations {tot / 1e6:.3f} ms")
for k in sorted(hist):
    print(f"# {k} kernels in flight: {hist[k] / 1e6:.3f} ms")
