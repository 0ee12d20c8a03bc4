#!/usr/bin/env python3
"""Per-workgroup record of one ping-pong convolution launch (csrc/conv_pp.hip; needs the profiling build: bash tools/build_prof.sh, then
CDET_LIB_PATH=tools/debug/_build/libcdet_prof.so). Prints prologue / loop / epilogue clocks per workgroup and, for wave 0 of group 0, the mean length
of a memory-phase body, of a memory phase up to its barrier release, and of a compute phase up to its barrier release (s_memtime clocks).
Usage: python tools/pp_timeline.py --custom 40,40,320,320 [--bs 32] [--mode silu|raw]"""
import argparse
import ctypes as C
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cerberusdet_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--custom", type=str, default="40,40,320,320", help="H,W,Cin,Cout (3x3)")
    ap.add_argument("--mode", default="silu", choices=["silu", "raw"])
    a = ap.parse_args()
    H, W, ci, co = (int(v) for v in a.custom.split(","))
    k = 3
    dev, dtype = "cuda", torch.bfloat16
    lib = L.load()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(a.bs, H, W, ci, generator=g, device=dev).to(dtype)
    w = torch.randn(co, ci, k, k, generator=g, device=dev) / math.sqrt(ci * k * k)
    scale = torch.rand(co, generator=g, device=dev) + 0.5
    bias = torch.randn(co, generator=g, device=dev) * 0.1
    src, y = ops.View(x), ops.new_act(a.bs, H, W, co, dtype)
    wt, _ = ops.pack_weight_tiled(w, dtype)
    if a.mode == "silu":
        run = lambda: ops.conv2d_tiled(src, wt, y, k, scale=scale, bias=bias, act=L.ACT_SILU)  # noqa: E731
    else:
        stats = torch.zeros(ops.conv_tiled_stat_blocks(src, y, k) * 2 * co, device=dev)
        run = lambda: ops.conv2d_tiled(src, wt, y, k, stats=stats)  # noqa: E731
    for _ in range(3):
        run()
    n_p = ops.conv_tiled_stat_blocks(src, y, k)
    nblk = ((n_p + 1) // 2) * ((co + 159) // 160)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    fn = lib.cdet_debug_pp_timeline
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
    assert fn(buf.data_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    fn(None)
    t = buf.cpu().numpy().reshape(nblk, 8).astype(np.int64)
    assert (t[:, 3] > 0).all(), "no records: not a ping-pong launch (or not the profiling build)"
    t0 = t[:, 0].min()
    nsteps = ((ci + 31) // 32) * 9
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"shape {(H, W, ci, co)} mode {a.mode}: {nblk} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented), span {(t[:, 3] - t0).max()} clocks")
    print(f"per workgroup (median clocks): prologue {np.median(pro):.0f}  loop {np.median(loop):.0f} ({np.median(loop) / nsteps:.0f} per K step = two phases)  epilogue {np.median(epi):.0f}")
    print(f"start spread: last workgroup starts {(t[:, 0] - t0).max()} clocks after the first; end spread {(t[:, 3].max() - t[:, 3].min())}")
    print(f"group 0 / wave 0, mean per phase: memory body {np.median(t[:, 4]) / nsteps:.0f}, memory phase to release {np.median(t[:, 5]) / nsteps:.0f}, "
          f"compute phase to release {np.median(t[:, 6]) / nsteps:.0f} clocks (640 = 20 MFMAs)")
    xcc = t[:, 7] & 0xf
    print("workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())


if __name__ == "__main__":
    main()
