#!/usr/bin/env python3
"""Cost of the eval-form epilogue of the tap-resident kernel: raw 16-bit output, + scale / bias / SiLU, + residual, per dominant shape (batch 32, bf16)."""
import math
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cerberusdet_amd import _lib as L, ops  # noqa: E402

SHAPES = [(160, 160, 80, 80), (80, 80, 160, 160), (40, 40, 320, 320), (80, 80, 320, 320), (20, 20, 320, 320)]


def main():
    dev, dtype, N = "cuda", torch.bfloat16, 32
    g = torch.Generator(device=dev).manual_seed(5)
    print(f"{'shape':22s} {'raw':>8s} {'silu':>8s} {'silu+res':>9s}   (ms)")
    for H, W, ci, co in SHAPES:
        x = torch.randn(N, H, W, ci, generator=g, device=dev).to(dtype)
        r = torch.randn(N, H, W, co, generator=g, device=dev).to(dtype)
        w = torch.randn(co, ci, 3, 3, generator=g, device=dev) / math.sqrt(ci * 9)
        sc, bi = torch.rand(co, generator=g, device=dev) + 0.5, torch.randn(co, generator=g, device=dev) * 0.1
        wt, _ = ops.pack_weight_tiled(w, dtype)
        y = ops.new_act(N, H, W, co, dtype)
        src, res = ops.View(x), ops.View(r)
        fns = {
            "raw": lambda: ops.conv2d_tiled(src, wt, y, 3),
            "silu": lambda: ops.conv2d_tiled(src, wt, y, 3, scale=sc, bias=bi, act=L.ACT_SILU),
            "silu+res": lambda: ops.conv2d_tiled(src, wt, y, 3, scale=sc, bias=bi, act=L.ACT_SILU, res=res),
        }
        times = {k: [] for k in fns}
        for fn in fns.values():
            fn()
        torch.cuda.synchronize()
        for _ in range(5):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 10)
        m = {k: statistics.median(v) for k, v in times.items()}
        print(f"{H:3d}x{W:<3d} {ci:4d}->{co:<4d}     {m['raw']:8.4f} {m['silu']:8.4f} {m['silu+res']:9.4f}")


if __name__ == "__main__":
    main()
