"""north-star forward timing trend after an idle period: consecutive 20-forward blocks (clock ramp vs stream state)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
model.eval().bfloat16()
x = torch.rand(32, 3, 640, 640).bfloat16().to(dev)


def blocks(tag, n=12, reps=20):
    out = []
    with torch.no_grad():
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                model(x)
            e1.record()
            torch.cuda.synchronize()
            out.append(round(e0.elapsed_time(e1) / reps, 2))
    print(tag, out, flush=True)


blocks("fresh")
time.sleep(5)
blocks("after 5 s idle")
time.sleep(0.5)
blocks("after 0.5 s idle")
# host busy (no GPU work) for 5 s
t = time.time()
while time.time() - t < 5:
    sum(range(10000))
blocks("after 5 s host-busy")
