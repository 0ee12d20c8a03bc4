#!/usr/bin/env python3
"""Weight gradient of ONE stride-1 shape at batch 32 (random bf16 operands), single or grouped launch: median ms and TF/s.
Usage: python tools/wgrad_bench.py --shape H,W,Cin,Cout,k[,layers] [--shape ...]   (CDET_LIB_PATH selects an ablation build)"""
import argparse
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cerberusdet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", action="append", default=[])
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dev, dtype = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(3)
    for t in a.shape or ["160,160,80,80,3,6"]:
        v = [int(x) for x in t.split(",")]
        H, W, ci, co, k = v[:5]
        n = v[5] if len(v) > 5 else 1
        items = []
        for _ in range(n):
            x = torch.randn(a.bs, H, W, ci, generator=g, device=dev).to(dtype)
            dy = torch.randn(a.bs, H, W, co, generator=g, device=dev).to(dtype)
            items.append((ops.View(x), ops.View(dy), torch.zeros(co, ci, k, k, device=dev)))
        if n > 1:
            fn = lambda: ops.conv2d_wgrad_grouped(items, k, 1)  # noqa: E731
        else:
            ws = torch.empty(1 << 26, device=dev)
            fn = lambda: ops.conv2d_wgrad(items[0][0], items[0][1], items[0][2], k, 1, ws=ws)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ms = statistics.median(ts)
        fl = 2.0 * a.bs * H * W * ci * co * k * k * n
        print(f"wgrad {H}x{W} {ci}->{co} {k}x{k} x{n}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TF/s")


if __name__ == "__main__":
    main()
