#!/usr/bin/env python3
"""Experiment: the eval forward of a batch as K concurrent part-batch pipelines (separate plans / buffers / streams). In eval form
(BatchNorm folded) the samples are independent, so the result is the same; the question is whether kernels of DIFFERENT layers in flight
at once fill the workgroup slots that the single-round / 1.5-round launches of one pipeline leave idle."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cerberusdet_amd import engine

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
model.eval()
tasks = list(bench.TASKS)
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timed(fn, n=40, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


x = torch.rand(BS, 3, 640, 640, device=dev).to(torch.bfloat16)
base = timed(lambda: model(x, zero_copy=True))
print(f"one pipeline, batch {BS}: {base:.3f} ms")
for K in (2, 4):
    n = BS // K
    xs = [x[i * n:(i + 1) * n].contiguous() for i in range(K)]
    plans, streams = [], [torch.cuda.Stream(device=dev) for _ in range(K)]
    orig = engine.lane_stream
    for i in range(K):
        p = engine.Plan(model, tasks, n, 640, 640, False, model.compute_dtype, x.dtype, dev)
        engine.lane_stream = lambda d, ln, i=i: orig(d, ln + 8 * (i + 1))  # every pipeline gets lane streams of its own
        with torch.cuda.stream(streams[i]):
            p.run_forward(xs[i])
        plans.append(p)
    engine.lane_stream = orig
    torch.cuda.synchronize()

    def run():
        cur = torch.cuda.current_stream()
        for i in range(K):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                plans[i].run_forward(xs[i])
        for i in range(K):
            cur.wait_stream(streams[i])

    t = timed(run)
    print(f"{K} pipelines of batch {n}: {t:.3f} ms  ({base / t:.3f}x)")
