#!/usr/bin/env python3
"""What the BatchNorm partial-sum epilogue and the strided source cost on the tap-resident kernel (train form: raw 16-bit output + statistics).
Per shape: raw output without / with statistics, from a dense source and from a channel slice of a wider (concat) buffer."""
import math
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cerberusdet_amd import ops  # noqa: E402

SHAPES = [(160, 160, 80, 80, 400), (80, 80, 160, 160, 800), (40, 40, 320, 320, 1600), (80, 80, 320, 320, 320), (20, 20, 320, 320, 1600)]


def main():
    dev, dtype, N = "cuda", torch.bfloat16, 32
    g = torch.Generator(device=dev).manual_seed(5)
    print(f"{'shape':24s} {'raw':>8s} {'raw+stats':>10s} {'slice src':>10s} {'slice+stats':>12s}   (ms; TF/s of the last)")
    for H, W, ci, co, ld in SHAPES:
        xd = torch.randn(N, H, W, ci, generator=g, device=dev).to(dtype)
        xw = torch.randn(N, H, W, ld, generator=g, device=dev).to(dtype)
        w = torch.randn(co, ci, 3, 3, generator=g, device=dev) / math.sqrt(ci * 9)
        wt, _ = ops.pack_weight_tiled(w, dtype)
        y = ops.new_act(N, H, W, co, dtype)
        srcs = {"dense": ops.View(xd), "slice": ops.View(xw, ci, ci) if ld >= 2 * ci else ops.View(xw, 0, ci)}
        nblk = ops.conv_tiled_stat_blocks(srcs["dense"], y, 3)
        st = torch.zeros(nblk * 2 * co, device=dev)
        fns = {
            "raw": lambda: ops.conv2d_tiled(srcs["dense"], wt, y, 3),
            "raw+stats": lambda: ops.conv2d_tiled(srcs["dense"], wt, y, 3, stats=st),
            "slice": lambda: ops.conv2d_tiled(srcs["slice"], wt, y, 3),
            "slice+stats": lambda: ops.conv2d_tiled(srcs["slice"], wt, y, 3, stats=st),
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
        fl = 2.0 * N * H * W * ci * co * 9
        print(f"{H:3d}x{W:<3d} {ci:4d}->{co:<4d} ld {ld:<5d} {m['raw']:8.4f} {m['raw+stats']:10.4f} {m['slice']:10.4f} {m['slice+stats']:12.4f}   {fl / m['slice+stats'] / 1e9:6.0f}")


if __name__ == "__main__":
    main()
