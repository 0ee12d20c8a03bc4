#!/usr/bin/env python3
"""Achieved HBM bandwidth of the BatchNorm/SiLU/pool kernels on the [M, C] shapes of one YOLOv8x task pass (bs 32 @640).
Algorithmic bytes: fwd = z read + y write (+ residual read); reduce = dy + z read; apply = dy + z read + dz write.
Usage: python tools/ew_bench.py [--reps 20]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cerberusdet_amd import ops  # noqa: E402
from cerberusdet_amd import _lib as L  # noqa: E402

# (N, H, W, C, launches per task pass) of the C2f bottleneck / cv1 / cv2 outputs that dominate the BN traffic
SHAPES = [(32, 160, 160, 80, 8), (32, 160, 160, 160, 3), (32, 80, 80, 160, 20), (32, 80, 80, 320, 4), (32, 40, 40, 320, 30),
          (32, 40, 40, 640, 5), (32, 20, 20, 320, 15), (32, 20, 20, 640, 6)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    lib = L.load()
    tot = dict(fwd=0.0, reduce=0.0, apply=0.0, finalize=0.0, sums=0.0)
    print(f"{'shape':>22s} {'fwd us':>8s} {'GB/s':>6s} {'reduce us':>9s} {'GB/s':>6s} {'apply us':>8s} {'GB/s':>6s} {'finalize us':>11s} {'sums us':>8s}")
    for N, H, W, Cn, cnt in SHAPES:
        z = ops.new_act(N, H, W, Cn, torch.bfloat16, dev)
        z.buf.copy_(torch.randn_like(z.buf, dtype=torch.float32))
        y = ops.new_act(N, H, W, Cn, torch.bfloat16, dev)
        dy = ops.new_act(N, H, W, Cn, torch.bfloat16, dev)
        dy.buf.copy_(torch.randn_like(dy.buf, dtype=torch.float32))
        dz = ops.new_act(N, H, W, Cn, torch.bfloat16, dev)
        mean = torch.zeros(Cn, device=dev)
        invstd = torch.ones(Cn, device=dev)
        gamma = torch.ones(Cn, device=dev)
        beta = torch.zeros(Cn, device=dev)
        dg, db = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev)
        M = N * H * W
        nb = lib.cdet_bn_bwd_blocks(M)
        part = torch.empty(nb * 2 * Cn + 2 * Cn, device=dev)
        nstat = (M + 127) // 128
        stats = torch.randn(nstat * 2 * Cn, device=dev)
        rm, rv = torch.zeros(Cn, device=dev), torch.ones(Cn, device=dev)
        st = ops.stream()
        P = ops.ptr
        dtc = ops.dt(torch.bfloat16)
        t_f = timeit(lambda: ops.bn_silu_fwd(z, mean, invstd, gamma, beta, y), a.reps)
        wide = ops.new_act(N, H, W, 5 * Cn, torch.bfloat16, dev)   # the same pass writing a channel slice of a concat buffer (pitch 5 C)
        ys = ops.View(wide.buf, Cn, Cn)
        t_fs = timeit(lambda: ops.bn_silu_fwd(z, mean, invstd, gamma, beta, ys), a.reps)
        del wide, ys
        t_r = timeit(lambda: lib.cdet_bn_silu_bwd_reduce(P(dy), dy.ld, 0, P(z), z.ld, 0, P(mean), P(invstd), P(gamma), P(beta), P(part), M, Cn, dtc,
                                                         st), a.reps)
        t_a = timeit(lambda: lib.cdet_bn_silu_bwd_apply(P(dy), dy.ld, 0, P(z), z.ld, 0, P(mean), P(invstd), P(gamma), P(beta), P(part), 0, None, None,
                                                        0, P(dz), dz.ld, 0, M, Cn, dtc, 0, st), a.reps)
        t_fin = timeit(lambda: ops.bn_finalize(stats, nstat, Cn, M, 1e-3, 0.03, rm, rv, mean, invstd), a.reps)
        t_s = timeit(lambda: lib.cdet_bn_bwd_sums(P(part), nb, Cn, P(part) + nb * 2 * Cn * 4, P(dg), P(db), 1, st), a.reps)
        by = M * Cn * 2
        print(f"{f'{N}x{H}x{W}x{Cn} (x{cnt})':>22s} {t_f*1e3:8.1f} {2*by/t_f/1e6:6.0f} {t_r*1e3:9.1f} {2*by/t_r/1e6:6.0f} {t_a*1e3:8.1f} "
              f"{3*by/t_a/1e6:6.0f} {t_fin*1e3:11.1f} {t_s*1e3:8.1f}   fwd->slice {t_fs*1e3:6.1f} us {2*by/t_fs/1e6:6.0f}")
        for k, t in (("fwd", t_f), ("reduce", t_r), ("apply", t_a), ("finalize", t_fin), ("sums", t_s)):
            tot[k] += t * cnt
    print("per task pass (ms): " + "  ".join(f"{k} {v:.2f}" for k, v in tot.items()))
    # SPPF pool backward
    buf = ops.new_act(32, 20, 20, 1280, torch.bfloat16, dev)
    buf.buf.copy_(torch.randn_like(buf.buf, dtype=torch.float32))
    ops.sppf_pool(buf, 320)
    dbuf = ops.new_act(32, 20, 20, 1280, torch.bfloat16, dev)
    dbuf.buf.copy_(torch.randn_like(dbuf.buf, dtype=torch.float32))
    t_p = timeit(lambda: ops.sppf_pool_bwd(buf, dbuf, 320), a.reps)
    print(f"sppf_pool_bwd 32x20x20x(4x320): {t_p*1e3:.1f} us for the 3 chained stages")


if __name__ == "__main__":
    main()
