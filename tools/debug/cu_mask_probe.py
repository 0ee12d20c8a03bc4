#!/usr/bin/env python3
"""Probe for a by-kernel-type CU partition (round 6): how fast do the MFMA-bound convolutions run on 192 / 208 / 224 of the 256 CUs, how fast the HBM-bound
BatchNorm passes on the complementary 64 / 48 / 32 -- each alone on its masked stream, then BOTH AT ONCE (convolution on the big mask, BatchNorm passes
on the small one). hipExtStreamCreateWithCUMask; masks interleaved (every k-th CU) so that all XCDs / SEs stay balanced."""
import ctypes as C
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cerberusdet_amd import _lib as L, ops  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(keep):
    ncu = 256
    bits = [0] * 8
    for i in range(ncu):
        if keep(i):
            bits[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), (C.c_uint32 * 8)(*bits))
    assert rc == 0 and st.value
    s = torch.cuda.ExternalStream(st.value)
    torch.cuda.Event().record(s)
    return s, sum(bin(b).count("1") for b in bits)


def timeit(fn, stream, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(5)
    convs = {}
    for H, ci, co in ((40, 320, 320), (80, 160, 160), (80, 320, 320), (160, 80, 80)):
        x = ops.View(torch.randn(32, H, H, ci, generator=g, device=dev).to(dt))
        w = torch.randn(co, ci, 3, 3, generator=g, device=dev) / math.sqrt(ci * 9)
        y = ops.new_act(32, H, H, co, dt)
        wt, _ = ops.pack_weight_tiled(w, dt)
        stats = torch.zeros(ops.conv_tiled_stat_blocks(x, y, 3) * 2 * co, device=dev)
        convs[(H, ci, co)] = (lambda x=x, wt=wt, y=y, stats=stats: ops.conv2d_tiled(x, wt, y, 3, stats=stats), 2.0 * 32 * H * H * ci * co * 9)
    bns = {}
    for H, cn in ((40, 320), (80, 160), (160, 80)):
        z = ops.new_act(32, H, H, cn, dt)
        z.buf.copy_(torch.randn_like(z.buf, dtype=torch.float32))
        yb = ops.new_act(32, H, H, cn, dt)
        mean, invstd, gamma, beta = torch.zeros(cn, device=dev), torch.ones(cn, device=dev), torch.ones(cn, device=dev), torch.zeros(cn, device=dev)
        bns[(H, cn)] = (lambda z=z, yb=yb, a=(mean, invstd, gamma, beta): ops.bn_silu_fwd(z, *a, yb), 2.0 * 32 * H * H * cn * 2)
    full = torch.cuda.current_stream()
    masks = {256: (full, 256)}
    for off_every, name in ((8, 224), (5, 205), (4, 192)):
        masks[name] = masked_stream(lambda i, k=off_every: i % k != k - 1)
    small = {}
    for off_every, name in ((8, 32), (5, 51), (4, 64)):
        small[name] = masked_stream(lambda i, k=off_every: i % k == k - 1)
    print("== convolutions (train form: raw output + BatchNorm partial sums), alone, us / TF/s by CU count")
    for key, (fn, fl) in convs.items():
        row = []
        for name, (st, n) in masks.items():
            t = timeit(fn, st)
            row.append(f"{n}: {t:7.1f} us {fl / t / 1e6:6.0f} TF/s")
        print(f"  {key}: " + " | ".join(row))
    print("== bn_silu_fwd, alone, us / GB/s by CU count")
    for key, (fn, by) in bns.items():
        row = []
        for name, (st, n) in list(small.items()) + [(256, (full, 256))]:
            t = timeit(fn, st)
            row.append(f"{n}: {t:7.1f} us {by / t / 1e3:6.0f} GB/s")
        print(f"  {key}: " + " | ".join(row))
    print("== both at once: 20 convolutions on the big mask while BatchNorm passes loop on the complementary small mask (conv us; BN passes completed per conv)")
    for (big, sm) in ((224, 32), (205, 51), (192, 64)):
        stc, nb = masks[big]
        sts, ns = small[sm]
        for ckey in ((40, 320, 320), (80, 160, 160), (80, 320, 320)):
            fn, fl = convs[ckey]
            bfn, by = bns[(ckey[0], ckey[2])]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps, breps = 20, 60
            with torch.cuda.stream(sts):
                b0.record()
                for _ in range(breps):
                    bfn()
                b1.record()
            with torch.cuda.stream(stc):
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
            torch.cuda.synchronize()
            tc, tb = e0.elapsed_time(e1) / reps * 1e3, b0.elapsed_time(b1) / breps * 1e3
            print(f"  conv {ckey} on {nb} CUs: {tc:7.1f} us ({fl / tc / 1e6:5.0f} TF/s) beside bn {ckey[0]}x{ckey[2]} on {ns} CUs: {tb:7.1f} us ({by / tb / 1e3:5.0f} GB/s)")
    print("== reference: both on UNMASKED streams at once (today's two-stream situation)")
    s2 = torch.cuda.Stream()
    for ckey in ((40, 320, 320), (80, 160, 160), (80, 320, 320)):
        fn, fl = convs[ckey]
        bfn, by = bns[(ckey[0], ckey[2])]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s2):
            b0.record()
            for _ in range(60):
                bfn()
            b1.record()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        tc, tb = e0.elapsed_time(e1) / 20 * 1e3, b0.elapsed_time(b1) / 60 * 1e3
        print(f"  conv {ckey}: {tc:7.1f} us ({fl / tc / 1e6:5.0f} TF/s) beside bn: {tb:7.1f} us ({by / tb / 1e3:5.0f} GB/s)")


if __name__ == "__main__":
    main()
