#!/usr/bin/env python3
"""A/B of the generic implicit-GEMM kernel (cdet_conv2d) and the tap-resident kernel (cdet_conv2d_tiled) on the dominant
stride-1 shapes of the YOLOv8x path at batch 32 @640 (eval-form epilogue: scale, bias, SiLU). Random bf16 data.
Interleaved rounds in one process; prints median ms and TF/s per shape. Usage: python tools/conv_tiled_bench.py [--bs 32]"""
import argparse
import math
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cerberusdet_amd import _lib as L, ops  # noqa: E402

SHAPES = [  # H, W, Cin, Cout, k, launches per 2-task all-heads forward
    (40, 40, 320, 320, 3, 38), (80, 80, 160, 160, 3, 24), (80, 80, 320, 320, 3, 4), (20, 20, 320, 320, 3, 20),
    (40, 40, 640, 320, 3, 2), (40, 40, 1600, 640, 1, 4), (80, 80, 960, 320, 1, 2), (80, 80, 800, 320, 1, 2),
    (20, 20, 1280, 640, 1, 3), (160, 160, 160, 160, 1, 1), (80, 80, 320, 320, 1, 3), (40, 40, 640, 640, 1, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", type=int, default=-1, help="index into SHAPES")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands: the same instruction stream at far lower switching power -- if the "
                    "rate jumps, the kernel runs against the power / clock limit, not against its own stalls (MI355X_MICROARCH.md, DVFS)")
    ap.add_argument("--shape", action="append", default=[], help="H,W,Cin,Cout,k (repeatable): custom shapes instead of the table")
    a = ap.parse_args()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    tot = {"generic": 0.0, "tiled": 0.0}
    totf = 0.0
    print(f"{'shape':28s} {'generic ms':>10s} {'TF/s':>7s} {'tiled ms':>10s} {'TF/s':>7s} {'speedup':>8s}")
    shapes = [tuple(int(v) for v in t.split(",")) + (1,) for t in a.shape] or (SHAPES if a.only < 0 else [SHAPES[a.only]])
    for H, W, ci, co, k, n in shapes:
        x = torch.randn(a.bs, H, W, ci, generator=g, device=dev).to(dtype)
        w = torch.randn(co, ci, k, k, generator=g, device=dev) / math.sqrt(ci * k * k)
        scale = torch.rand(co, generator=g, device=dev) + 0.5
        bias = torch.randn(co, generator=g, device=dev) * 0.1
        if a.zeros:
            x.zero_()
            w.zero_()
        src = ops.View(x)
        y0, y1 = ops.new_act(a.bs, H, W, co, dtype), ops.new_act(a.bs, H, W, co, dtype)
        wp = ops.pack_weight(w, dtype)
        wt, _ = ops.pack_weight_tiled(w, dtype)
        fns = {
            "generic": lambda: ops.conv2d(src, wp, y0, k, 1, scale=scale, bias=bias, act=L.ACT_SILU),
            "tiled": lambda: ops.conv2d_tiled(src, wt, y1, k, scale=scale, bias=bias, act=L.ACT_SILU),
        }
        times = {kk: [] for kk in fns}
        for kk, fn in fns.items():
            fn()
        torch.cuda.synchronize()
        err = (y0.torch().float() - y1.torch().float()).abs().max().item()
        for _ in range(a.rounds):
            for kk, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[kk].append(e0.elapsed_time(e1) / a.reps)
        flops = 2.0 * a.bs * H * W * ci * co * k * k
        mg, mt = statistics.median(times["generic"]), statistics.median(times["tiled"])
        tot["generic"] += n * mg
        tot["tiled"] += n * mt
        totf += n * flops
        print(f"{H:3d}x{W:<3d} {ci:4d}->{co:<4d} {k}x{k} x{n:<3d}   {mg:10.4f} {flops / mg / 1e9:7.0f} {mt:10.4f} {flops / mt / 1e9:7.0f} {mg / mt:8.2f}  maxdiff {err:.3g}")
    for kk in tot:
        print(f"TOTAL {kk}: {tot[kk]:.2f} ms, {totf / tot[kk] / 1e9:.0f} TF/s over the listed launches")


if __name__ == "__main__":
    main()
