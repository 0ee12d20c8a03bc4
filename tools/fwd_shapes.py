#!/usr/bin/env python3
"""Per-launch timing of the north-star forward (YOLOv8x 2-task all-heads, eval form, bf16, batch 32 @640): every call of the compiled
launch list, grouped by (entry point, shape), HIP-event timed alone on the GPU. Usage: python tools/fwd_shapes.py [--bs 32] [--half]"""
import argparse
import sys
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--half", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model("v8x_2task.yaml", dev)
    model.eval()
    model = model.half() if a.half else model.bfloat16()
    x = torch.rand(a.bs, 3, 640, 640, generator=torch.Generator().manual_seed(3))
    x = (x.half() if a.half else x.bfloat16()).to(dev)
    with torch.no_grad():
        for _ in range(2):
            model(x)
    torch.cuda.synchronize()
    plan = model.get_plan(bench.TASKS, x.shape, x.dtype, training=False)
    groups = OrderedDict()
    for fn, args in plan.fwd:
        name = getattr(fn, "__name__", "fn")
        flops = 0.0
        key = (name,)
        if name.startswith("cdet_conv2d"):
            d = args[0]._obj
            key = (name, d.Hd, d.Wd, d.Cs, d.Cd, d.kh, d.stride, d.out_dtype)
            flops = 2.0 * d.N * d.Hd * d.Wd * d.Cd * d.Cs * d.kh * d.kw
        g = groups.setdefault(key, dict(n=0, flops=flops, call=(fn, args)))
        g["n"] += 1
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for key, g in groups.items():
        fn, args = g["call"]
        for _ in range(2):
            fn(*args, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn(*args, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        rows.append((key, g["n"], ms, g["flops"]))
    tot = sum(n * ms for _, n, ms, _ in rows)
    print(f"{'entry':26s} {'HxW':>9s} {'Cin':>5s} {'Cout':>5s} k s {'n':>3s} {'ms':>8s} {'TF/s':>7s} {'n*ms':>8s} {'%':>5s}")
    for key, n, ms, fl in sorted(rows, key=lambda r: -r[1] * r[2]):
        if len(key) > 1:
            name, H, W, ci, co, k, s, od = key
            print(f"{name:26s} {H:4d}x{W:<4d} {ci:5d} {co:5d} {k} {s} {n:3d} {ms:8.4f} {fl / ms / 1e9:7.0f} {n * ms:8.3f} {100 * n * ms / tot:5.1f}" + (" f32out" if od == 2 else ""))
        else:
            print(f"{key[0]:26s} {'':9s} {'':5s} {'':5s}     {n:3d} {ms:8.4f} {'':7s} {n * ms:8.3f} {100 * n * ms / tot:5.1f}")
    fl = sum(n * f for _, n, _, f in rows)
    print(f"TOTAL {tot:.3f} ms sequential, {len(plan.fwd)} launches, {fl / 1e12:.2f} TFLOP counted -> {fl / tot / 1e9:.0f} TF/s")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        for _ in range(10):
            model(x)
        e1.record()
    torch.cuda.synchronize()
    print(f"model(x): {e0.elapsed_time(e1) / 10:.3f} ms per forward")


if __name__ == "__main__":
    main()
