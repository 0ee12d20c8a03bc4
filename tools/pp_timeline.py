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
    buf = torch.zeros(nblk * 8 + 800, dtype=torch.int64, device=dev)
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
    raw = buf.cpu().numpy().astype(np.int64)
    steps = raw[nblk * 8:].reshape(2, 400)
    t = raw[:nblk * 8].reshape(nblk, 8)
    assert (t[:, 3] > 0).all(), "no records: not a ping-pong launch (or not the profiling build)"
    xcc = t[:, 7] & 0xf
    nsteps = ((ci + 31) // 32) * 9
    # s_memtime counters of different XCCs are not synchronised: every time is taken relative to the first start on the workgroup's own XCC
    t0 = np.zeros(nblk, dtype=np.int64)
    for xc in range(8):
        m = xcc == xc
        if m.any():
            t0[m] = t[m, 0].min()
    st, lo, ep, en = (t[:, i] - t0 for i in range(4))
    pro, loop, epi = lo - st, ep - lo, en - ep
    print(f"shape {(H, W, ci, co)} mode {a.mode}: {nblk} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (event-timed, this build), span per XCC (max end) {[int(en[xcc == xc].max()) for xc in range(8) if (xcc == xc).any()]}")
    print(f"per workgroup clocks  (min / median / max): start {st.min()}/{int(np.median(st))}/{st.max()}  prologue {pro.min()}/{int(np.median(pro))}/{pro.max()}  "
          f"loop {loop.min()}/{int(np.median(loop))}/{loop.max()} ({np.median(loop) / nsteps:.0f} per K step)  epilogue {epi.min()}/{int(np.median(epi))}/{epi.max()}  end {en.min()}/{int(np.median(en))}/{en.max()}")
    if t[:, 6].max() == 0:  # no per-phase stamps: slots 4 / 5 hold s_memrealtime (100 MHz, chip-wide) at the workgroup's start and end
        r0 = t[:, 4].min()
        rs, re_ = (t[:, 4] - r0) / 100.0, (t[:, 5] - r0) / 100.0
        print(f"chip-wide clock (us after the first workgroup's start): starts min/median/max {rs.min():.2f}/{np.median(rs):.2f}/{rs.max():.2f}   "
              f"ends {re_.min():.2f}/{np.median(re_):.2f}/{re_.max():.2f}   workgroup life median {np.median(re_ - rs):.2f} us = {np.median(en - st) / np.median(re_ - rs) / 1e3:.2f} GHz of s_memtime")
        order = np.argsort(rs)
        print("  start times (us) of every 20th workgroup in start order:", [round(float(rs[i]), 2) for i in order[::20]])
    else:
        print(f"group 0 / wave 0, mean per phase: memory body {np.median(t[:, 4]) / nsteps:.0f}, memory phase to release {np.median(t[:, 5]) / nsteps:.0f}, "
              f"compute phase to release {np.median(t[:, 6]) / nsteps:.0f} clocks (640 = 20 MFMAs; the stamps stretch the phases)")
    if steps[0, 1] > 0:  # CDET_PP_ABLATE=64: per-step stamps of workgroup 0 (start of every compute phase, wave 0 of group 0 / group 1)
        for gi in (0, 1):
            d = np.diff(steps[gi, :nsteps])
            print(f"workgroup 0, group {gi}: clocks per K step (compute-phase start to compute-phase start): first 12 {d[:12].tolist()}  median {int(np.median(d))}  "
                  f"last 6 {d[-6:].tolist()}  max {int(d.max())} at step {int(d.argmax())}; every 9th (chunk boundaries) median {int(np.median(d[8::9]))}")
        print(f"  group 1 lags group 0 by (clocks) first / median / last: {int(steps[1, 0] - steps[0, 0])} / {int(np.median(steps[1, :nsteps] - steps[0, :nsteps]))} / {int(steps[1, nsteps - 1] - steps[0, nsteps - 1])}")
        print(f"  loop start -> first compute phase {int(steps[0, 0] - t[0, 1])} clocks; last compute-phase start -> epilogue stamp {int(t[0, 2] - steps[0, nsteps - 1])}")
    print("workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())


if __name__ == "__main__":
    main()
