#!/usr/bin/env python3
"""Latency of the 2-task all-heads forward at small batch: wall per call (host enqueue + GPU) vs GPU time under HIP events."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
model.eval()
for bs in (1, 2, 4, 8, 16, 32):
    x = torch.rand(bs, 3, 640, 640, device=dev).to(torch.bfloat16)
    for _ in range(10):
        model(x, zero_copy=True)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        model(x, zero_copy=True)
    t_host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / n * 1e3
    lat = []
    for _ in range(20):
        t1 = time.perf_counter()
        model(x, zero_copy=True)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat.sort()
    print(f"bs {bs:3d}: enqueue {t_host:6.2f} ms/call, back-to-back {t_wall:6.2f} ms/call ({bs / t_wall * 1e3:7.1f} img/s), single-call latency p50 {lat[10]:6.2f} ms")
