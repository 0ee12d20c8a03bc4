"""north-star forward time vs buffer placement: the eval plan is rebuilt several times in one process (with allocator churn in
between); CDET_ARENA / CDET_ARENA_ALIGN / CDET_ARENA_SKEW select the placement policy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
x = torch.rand(32, 3, 640, 640).bfloat16().to(dev)
model.eval().bfloat16()


def blocks(tag, n=4, reps=20):
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
    print(tag, out[1:], flush=True)


junk = []
for i in range(5):
    model._plans = {}
    blocks(f"build {i}")
    # allocator churn: odd-sized blocks allocated and partly freed, so the next build lands elsewhere
    junk.append(torch.empty((i + 1) * 37_000_001, dtype=torch.uint8, device=dev))
    tmp = [torch.empty(11_000_003 * (j + 1), dtype=torch.uint8, device=dev) for j in range(6)]
    del tmp
