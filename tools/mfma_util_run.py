#!/usr/bin/env python3
"""Launches the five dominant MFMA kernel instantiations of the YOLOv8x step at batch 32 @640, each `--reps` times, on random or all-zero
(--zeros) operands -- the workload tools/pmc_mfma.sh runs under rocprofv3 counter passes (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, ...).
  conv_halo_kernel   linear tiles   40 x 40 320 -> 320 3x3   (38 launches per north-star forward, the most frequent)
  conv_halo_kernel   16 x 16 patch  80 x 80 160 -> 160 3x3
  conv_pair_kernel                  40 x 40 1600 -> 640 1x1
  conv_halo_kernel   96-cout patch  160 x 160 80 -> 80 3x3   (round 5: three workgroups per CU; CDET_HALO_WG3=0 for two)
  conv_vt_kernel     stride 2       160 x 160 160 -> 320 3x3 (80 x 80 output)
  wgrad_halo_kernel                 40 x 40 320 -> 320 3x3 weight gradient"""
import argparse
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cerberusdet_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--bs", type=int, default=32)
    a = ap.parse_args()
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(5)

    def act(N, H, W, C):
        t = torch.randn(N, H, W, C, generator=g, device=dev)
        return ops.View((t * 0 if a.zeros else t).to(dt))

    def wt(Co, Ci, k):
        w = torch.randn(Co, Ci, k, k, generator=g, device=dev) / math.sqrt(Ci * k * k)
        return w * 0 if a.zeros else w

    N = a.bs
    jobs = []
    # round 6: the 40 x 40 layer twice -- on the 8-wave ping-pong form (csrc/conv_pp.hip, the default for single-round grids) and, pinned through the
    # library's switch, on the 4-wave form it replaced -- and the 40 x 40 640 -> 320 layer (twice the K loop) on the ping-pong form
    if True:
        x, w = act(N, 40, 40, 320), wt(320, 320, 3)
        y = ops.new_act(N, 40, 40, 320, dt)
        wf, _ = ops.pack_weight_tiled(w, dt)
        sc, bi = torch.ones(320, device=dev), torch.zeros(320, device=dev)

        def four_wave(x=x, wf=wf, y=y, sc=sc, bi=bi):
            L.set_switch("conv_pp", 0)
            ops.conv2d_tiled(x, wf, y, 3, scale=sc, bias=bi, act=L.ACT_SILU)
            L.set_switch("conv_pp", None)
        jobs.append(four_wave)
    for H, Ci, Co, k in ((40, 640, 320, 3), (40, 320, 320, 3), (80, 160, 160, 3), (40, 1600, 640, 1), (160, 80, 80, 3)):  # (the summary keeps the LAST launches of a kernel name: the 320 -> 320 layer for conv_pp_kernel)
        x, w = act(N, H, H, Ci), wt(Co, Ci, k)
        y = ops.new_act(N, H, H, Co, dt)
        wf, _ = ops.pack_weight_tiled(w, dt)
        sc, bi = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
        jobs.append(lambda x=x, wf=wf, y=y, k=k, sc=sc, bi=bi: ops.conv2d_tiled(x, wf, y, k, scale=sc, bias=bi, act=L.ACT_SILU))
    x, w = act(N, 160, 160, 160), wt(320, 160, 3)
    y = ops.new_act(N, 80, 80, 320, dt)
    wf, _ = ops.pack_weight_tiled(w, dt)
    sc, bi = torch.ones(320, device=dev), torch.zeros(320, device=dev)
    jobs.append(lambda: ops.conv2d_s2_tiled(x, wf, y, scale=sc, bias=bi, act=L.ACT_SILU))
    xw, dyw = act(N, 40, 40, 320), act(N, 40, 40, 320)
    dw = torch.zeros(320, 320, 3, 3, device=dev)
    jobs.append(lambda: ops.conv2d_wgrad(xw, dyw, dw, 3, 1))
    for j in jobs:
        for _ in range(2):
            j()
    torch.cuda.synchronize()
    for j in jobs:
        for _ in range(a.reps):
            j()
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
