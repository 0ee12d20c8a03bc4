#!/usr/bin/env python3
"""Per-shape timing of every distinct convolution launch of one YOLOv8x task pass (forward, dgrad, wgrad) at bs 32 @640.
Replays the plan's own pre-built calls (valid buffers) under HIP events. Usage: python tools/conv_shapes.py [--bs 32] [--reps 10]"""
import argparse
import sys
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from cerberusdet_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--only", default="", help="filter: HxW-Cin-Cout-k, e.g. 40x40-320-320-3")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model("v8x_2task.yaml", dev)
    from cerberusdet_amd.trainers import Averaging

    tr = Averaging(dev, model, bench.HYP, bench.TASKS, use_ema=False)
    b = bench.synth_batch(0, 0, 0, a.bs, 20, 640, dev)
    tr.forward_backward("voc", b, n_max=8, active_tasks=bench.TASKS)
    torch.cuda.synchronize()
    plan = model.get_plan("voc", b["img"].shape, b["img"].dtype, training=True)
    calls = [(fn, args) for fn, args in plan.fwd] + [(fn, args) for _, cs in plan.bwd_groups for fn, args in cs]
    groups = OrderedDict()
    for fn, args in calls:
        name = getattr(fn, "__name__", "")
        if name not in ("cdet_conv2d", "cdet_conv2d_wgrad", "cdet_conv2d_wgrad_grouped", "cdet_conv2d_tiled", "cdet_conv2d_tiled_bn", "cdet_conv2d_tiled_dgrad",
                        "cdet_conv2d_s2_tiled", "cdet_conv2d_s2_tiled_bn", "cdet_conv2d_s2_tiled_dgrad"):
            continue
        d = args[0]._obj
        if name == "cdet_conv2d_wgrad_grouped":  # n = layers; the timed call is the whole group (ms and TF/s are per group launch)
            key = ("wgrad", d.Hd, d.Wd, d.Cs, d.Cd, d.kh, -args[2])
            g = groups.setdefault(key + (len(groups),), dict(n=0, flops=args[2] * 2.0 * d.N * d.Hd * d.Wd * d.Cd * d.Cs * 9, call=(fn, args)))
            g["n"] += 1
            continue
        kind = "wgrad" if name == "cdet_conv2d_wgrad" else ("dgrad" if (d.mode == L.CONV_DGRAD or name.endswith("dgrad")) else "fwd")
        if name == "cdet_conv2d_s2_tiled_dgrad":
            key = (kind, d.Hd, d.Wd, d.Cd, d.Cs, d.kh, 2)
            flops = 2.0 * d.N * d.Hs * d.Ws * d.Cs * d.Cd * d.kh * d.kw
        elif name.endswith("tiled_dgrad"):
            key = (kind, d.Hd, d.Wd, d.Cd, d.Cs, d.kh, 1)
            flops = 2.0 * d.N * d.Hs * d.Ws * d.Cs * d.Cd * d.kh * d.kw
        elif kind == "dgrad":
            key = (kind, d.Hd, d.Wd, d.Cd, d.Cs, d.kh, d.stride)  # dX spatial, Cin=Cd, Cout=Cs
            flops = 2.0 * d.N * d.Hs * d.Ws * d.Cs * d.Cd * d.kh * d.kw
        else:
            key = (kind, d.Hd, d.Wd, d.Cs, d.Cd, d.kh, d.stride)
            flops = 2.0 * d.N * d.Hd * d.Wd * d.Cd * d.Cs * d.kh * d.kw
        g = groups.setdefault(key, dict(n=0, flops=flops, call=(fn, args)))
        g["n"] += 1
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for key, g in groups.items():
        if key[0] not in a.what.split(","):
            continue
        if a.only and a.only != f"{key[1]}x{key[2]}-{key[3]}-{key[4]}-{key[5]}":  # (grouped launches: stride column = -layers)
            continue
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
        rows.append((key, g["n"], ms, g["flops"] / ms / 1e9))
    tot = {}
    print(f"{'kind':6s} {'HxW(out)':>9s} {'Cin':>5s} {'Cout':>5s} k s {'n':>3s} {'ms':>8s} {'TF/s':>7s} {'n*ms':>8s}")
    for key, n, ms, tf in sorted(rows, key=lambda r: (r[0][0], -r[1] * r[2])):
        kind, H, W, ci, co, k, s = key[:7]
        print(f"{kind:6s} {H:4d}x{W:<4d} {ci:5d} {co:5d} {k} {s} {n:3d} {ms:8.3f} {tf:7.1f} {n * ms:8.2f}")
        t = tot.setdefault(kind, [0.0, 0.0])
        t[0] += n * ms
        t[1] += n * ms * tf
    for kind, (ms, w) in tot.items():
        print(f"TOTAL {kind}: {ms:.2f} ms per task pass, {w / ms:.1f} TF/s")


if __name__ == "__main__":
    main()
