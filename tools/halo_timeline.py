#!/usr/bin/env python3
"""Per-workgroup timeline of one cdet_conv2d_tiled launch (needs a -DCDET_PROFILING build: make -C cerberusdet_amd/csrc EXTRA=-DCDET_PROFILING).
Prints how many workgroups share a CU at the same time and the duration of prologue / K loop / epilogue.
Usage: python tools/halo_timeline.py [--shape 0] [--bs 32]"""
import argparse
import ctypes as C
import math
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cerberusdet_amd import _lib as L, ops  # noqa: E402
from tools.conv_tiled_bench import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--custom", type=str, default="", help="H,W,Cin,Cout,k instead of --shape")
    ap.add_argument("--mode", default="silu", choices=["silu", "raw"], help="eval epilogue (scale/bias/SiLU) or train form (raw output + BN partial sums)")
    a = ap.parse_args()
    H, W, ci, co, k, _ = SHAPES[a.shape] if not a.custom else tuple(int(v) for v in a.custom.split(",")) + (1,)
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
    hp = 256 * ops.conv_tiled_stat_blocks(src, y, k) // ((a.bs * H * W + 255) // 256) if False else None
    n_p = ops.conv_tiled_stat_blocks(src, y, k)  # pixel tiles (256-pixel, 16 x 16 patch or 128-pixel half tiles)
    rb = 96 if co <= 96 else 160
    nblk = n_p * ((co + rb - 1) // rb)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    fn = lib.cdet_debug_halo_timeline
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
    assert fn(buf.data_ptr()) == 0
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    fn(None)
    t = buf.cpu().numpy().reshape(nblk, 8)
    xcc, hw = t[:, 0] & 0xf, t[:, 1]
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    t = t.astype(np.int64)
    t0 = t[:, 2].min()
    st, lo, ep, en = (t[:, i] - t0 for i in (2, 3, 4, 5))
    print(f"shape {(H, W, ci, co, k)} mode {a.mode}, {nblk} workgroups on {len(set(cu.tolist()))} distinct CUs; kernel span {en.max()} clocks (s_memtime)")
    print(f"per workgroup (clocks): prologue {np.median(lo - st):.0f}  K loop {np.median(ep - lo):.0f}  epilogue {np.median(en - ep):.0f}   (min/max loop {(ep - lo).min()}/{(ep - lo).max()})")
    lp = (ep - lo).astype(np.float64)
    print(f"inside the K loop (wave 0): counted vmcnt/lgkmcnt wait {np.median(t[:, 6] / lp):.3f} of the loop, barrier behind it {np.median(t[:, 7] / lp):.3f}"
          f"  (absolute medians {np.median(t[:, 6]):.0f} / {np.median(t[:, 7]):.0f} clocks); lgkmcnt(0) alone {np.median(t[:, 1] / lp):.3f}")
    per = defaultdict(list)
    for i in range(nblk):
        per[int(cu[i])].append((int(st[i]), int(en[i])))
    conc = defaultdict(int)
    for c, iv in per.items():
        ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
        cur, last = 0, 0
        for tt, d in ev:
            conc[cur] += tt - last
            cur += d
            last = tt
    tot = sum(conc.values())
    print("time share by number of co-resident workgroups per CU:", {kk: round(v / tot, 3) for kk, v in sorted(conc.items())})
    print("workgroups per CU histogram:", dict(sorted({n: sum(1 for v in per.values() if len(v) == n) for n in set(len(v) for v in per.values())}.items())))
    first = sorted(per.items())[0][1]
    print("one CU's workgroups (start, loop, epi, end):", sorted((int(st[i]), int(lo[i]), int(ep[i]), int(en[i])) for i in range(nblk) if int(cu[i]) == sorted(per.items())[0][0]))


if __name__ == "__main__":
    main()
