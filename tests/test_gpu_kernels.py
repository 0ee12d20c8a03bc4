"""Kernel-level parity (MI355X): every C-ABI entry point against a CPU fp32 reference / the oracle.

Inputs are made exactly representable in the storage dtype first, so the only differences are the fp32 accumulation
order and the single output rounding: tolerance 2^-8 relative (one bf16 ulp) for bf16 outputs, 1e-4 for fp32 outputs.
Integer outputs (label assignment, NMS rows) are compared bit-exactly.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
from util import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from cerberusdet_amd import ops

    return ops


def _rt(x, dtype):  # round-trip through the storage dtype
    return x.to(dtype).float()


def _close(a, b, rtol, atol):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    bad = err > lim
    assert not bad.any(), f"max err {err.max():.4g} (ref max {b.abs().max():.4g}), {int(bad.sum())}/{bad.numel()} out of tolerance"


CONV_CASES = [
    # N, H, W, Cin, Cout, k, s, dtype
    (2, 20, 20, 160, 320, 3, 1, torch.bfloat16),   # wide tile, K = 1440 (22.5 K-steps -> zero-padded tail)
    (1, 24, 20, 80, 80, 3, 1, torch.bfloat16),     # narrow tile, Cin = 80 (taps straddle K-steps)
    (2, 16, 16, 320, 160, 1, 1, torch.bfloat16),   # 1x1
    (1, 22, 18, 80, 160, 3, 2, torch.bfloat16),    # stride 2, odd-ish sizes (M not a tile multiple)
    (1, 8, 8, 16, 24, 3, 1, torch.bfloat16),       # tiny channels (v8n-like), Cout not a multiple of 16
    (1, 12, 12, 400, 320, 1, 1, torch.float16),    # fp16, K = 400
    (3, 10, 10, 640, 640, 3, 2, torch.bfloat16),   # large K
    (2, 21, 17, 80, 160, 3, 2, torch.bfloat16),    # stride 2 on odd maps: the four dX parity classes have different sizes
    (1, 14, 14, 64, 48, 1, 2, torch.bfloat16),     # 1x1 stride 2: three of the four dX parity classes receive no tap
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bias_silu_residual_stats(case):
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, H, W, Ci, Co, k, s, dtype = case
    g = torch.Generator().manual_seed(1)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype)
    w = _rt(torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k), dtype)
    scale = torch.rand(Co, generator=g) + 0.5
    bias = torch.randn(Co, generator=g) * 0.1
    ref_raw = F.conv2d(x, w, None, s, k // 2)
    Ho, Wo = ref_raw.shape[2:]
    res = _rt(torch.randn(N, Co, Ho, Wo, generator=g), dtype)
    # source lives in a wider buffer at a channel offset, destination too (concat-slice semantics)
    xb = torch.zeros(N, H, W, Ci + 16, dtype=dtype, device=DEV)
    xb[..., 8:8 + Ci] = x.permute(0, 2, 3, 1).to(dtype).to(DEV)
    src = ops.View(xb, 8, Ci)
    wp = ops.pack_weight(w.to(DEV), dtype)
    # raw fp32 output + BN statistics
    dst32 = ops.new_act(N, Ho, Wo, Co, torch.float32)
    nblk = ops.conv_stat_blocks(src, dst32, k, s)
    stats = torch.zeros(nblk * 2 * Co, device=DEV)
    ops.conv2d(src, wp, dst32, k, s, stats=stats)
    torch.cuda.synchronize()
    _close(dst32.nchw(), ref_raw, 1e-4, 1e-4)
    st = stats.view(nblk, 2, Co).sum(0).cpu()
    _close(st[0], ref_raw.sum((0, 2, 3)), 1e-3, 1e-2)
    _close(st[1], (ref_raw ** 2).sum((0, 2, 3)), 1e-3, 1e-2)
    # fused epilogue into a slice of a wider buffer
    yb = torch.full((N, Ho, Wo, Co + 8), 7.0, dtype=dtype, device=DEV)
    dst = ops.View(yb, 8, Co)
    rv = ops.from_nchw(res.to(DEV), dtype)
    ops.conv2d(src, wp, dst, k, s, scale=scale.to(DEV), bias=bias.to(DEV), act=L.ACT_SILU, res=rv)
    torch.cuda.synchronize()
    ref = F.silu(ref_raw * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)) + res
    _close(dst.nchw(), ref, 2 ** -7, 2e-2)
    assert (yb[..., :8].float() == 7.0).all(), "conv wrote outside its channel slice"


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_and_wgrad(case):
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, H, W, Ci, Co, k, s, dtype = case
    Cop = (Co + 7) // 8 * 8
    g = torch.Generator().manual_seed(2)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype).requires_grad_(True)
    w = _rt(torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, s, k // 2)
    dy = _rt(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    dyv = ops.new_act(N, Ho, Wo, Cop, dtype, zero=True)
    dyv.buf[..., :Co] = dy.permute(0, 2, 3, 1).to(dtype).to(DEV)
    # dgrad: source = dy (Cs = Cop incl. zero pad), dest = dx
    wt = ops.pack_weight(w.detach().to(DEV), dtype, transpose=True, o_pad=Cop)
    dx = ops.new_act(N, H, W, Ci, torch.float32)
    ops.conv2d(dyv, wt, dx, k, s, mode=L.CONV_DGRAD)
    torch.cuda.synchronize()
    _close(dx.nchw(), x.grad, 1e-3, 1e-3)
    # accumulate into fp32
    ops.conv2d(dyv, wt, dx, k, s, mode=L.CONV_DGRAD, accumulate=True)
    torch.cuda.synchronize()
    _close(dx.nchw(), 2 * x.grad, 1e-3, 2e-3)
    # wgrad
    xv = ops.from_nchw(x.detach().to(DEV), dtype)
    dw = torch.zeros(Co, Ci, k, k, device=DEV)
    ops.conv2d_wgrad(xv, dyv, dw, k, s)
    torch.cuda.synchronize()
    _close(dw, w.grad, 2e-3, 2e-3 * float(w.grad.abs().max()))
    ops.conv2d_wgrad(xv, dyv, dw, k, s, accumulate=True)
    torch.cuda.synchronize()
    _close(dw, 2 * w.grad, 2e-3, 4e-3 * float(w.grad.abs().max()))


WGRAD_HALO_CASES = [
    # N, H, W, Cin, Cout, dtype -- 3x3 stride-1 shapes the tap-resident weight-gradient kernel (csrc/conv_wgrad_halo.hip) takes
    (2, 40, 40, 64, 160, torch.bfloat16),    # 4 pixel splits, the last one partial
    (1, 80, 80, 32, 136, torch.bfloat16),    # Cout not a multiple of 16: the last cout group is half empty
    (3, 24, 56, 96, 320, torch.float16),     # non-square map, two cout blocks, fp16
    (1, 16, 16, 32, 128, torch.bfloat16),    # one split, two stages
    (2, 20, 20, 128, 160, torch.bfloat16),   # Cin % 64 == 0: also runs on the 64-cin tile
    (1, 40, 40, 64, 200, torch.float16),     # 64-cin tile, two cout blocks, the second 40 wide
    (2, 40, 40, 80, 80, torch.bfloat16),     # 80-cout tile, Cin = 80: the last 32-cin tile is half empty
    (1, 32, 160, 80, 80, torch.bfloat16),    # 160-wide rows (the C2f 160 x 160 Bottlenecks): halo 450 rows
    (1, 20, 20, 640, 80, torch.float16),     # head box-branch stem
    (2, 24, 24, 96, 72, torch.bfloat16),     # Cout 72
    (2, 20, 100, 32, 192, torch.bfloat16),   # wide rows: the halo is 330 rows
]


@pytest.mark.parametrize("tile", ["32", "64"])
@pytest.mark.parametrize("case", WGRAD_HALO_CASES)
def test_conv_wgrad_tap_resident(case, tile, sw):
    """The tap-resident weight gradient against autograd AND against the im2col kernel it replaces on the same operands (channel
    slices of wider buffers on both sides)."""
    ops = _ops()
    N, H, W, Ci, Co, dtype = case
    Cop = (Co + 7) // 8 * 8
    g = torch.Generator().manual_seed(5)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype).requires_grad_(True)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1)
    dy = _rt(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    xb = torch.randn(N, H, W, Ci + 24, generator=g).to(dtype).to(DEV)
    xb[..., 16:16 + Ci] = x.detach().permute(0, 2, 3, 1).to(dtype).to(DEV)
    xv = ops.View(xb, 16, Ci)
    yb = torch.randn(N, H, W, Cop + 8, generator=g).to(dtype).to(DEV)
    yb[..., 8:8 + Cop] = 0
    yb[..., 8:8 + Co] = dy.permute(0, 2, 3, 1).to(dtype).to(DEV)
    dyv = ops.View(yb, 8, Cop)
    if tile == "64":
        if Ci % 64:
            pytest.skip("the 64-cin tile needs Cin % 64 == 0")
        sw("CDET_WGRAD_HALO", int("4"))
    dw = torch.zeros(Co, Ci, 3, 3, device=DEV)
    ops.conv2d_wgrad(xv, dyv, dw, 3, 1)
    torch.cuda.synchronize()
    _close(dw, w.grad, 2e-3, 2e-3 * float(w.grad.abs().max()))
    sw("CDET_WGRAD_HALO", int("0"))
    dw0 = torch.zeros(Co, Ci, 3, 3, device=DEV)
    ops.conv2d_wgrad(xv, dyv, dw0, 3, 1)
    torch.cuda.synchronize()
    sw("CDET_WGRAD_HALO", int("4" if tile == "64" else "1"))
    _close(dw, dw0, 1e-4, 1e-4 * float(w.grad.abs().max()))  # same bf16 products, fp32 sums in a different order
    ops.conv2d_wgrad(xv, dyv, dw, 3, 1, accumulate=True)
    torch.cuda.synchronize()
    _close(dw, 2 * w.grad, 2e-3, 4e-3 * float(w.grad.abs().max()))


WGRAD_PATCH_CASES = [
    # N, H, W, Cin, Cout, dtype -- 80-cout tile on maps of whole 8 x 16 patches: the patch form of csrc/conv_wgrad_halo.hip (four stage buffers)
    (1, 32, 160, 80, 80, torch.bfloat16),    # the C2f 160 x 160 Bottleneck rows: 40 patches, Cin 80 (the last 32-cin tile half empty)
    (2, 16, 48, 80, 80, torch.bfloat16),     # 12 patches over two images
    (1, 80, 80, 320, 80, torch.float16),     # head box-branch stem at 80 x 80: 10 cin tiles, fp16
    (3, 8, 16, 96, 72, torch.bfloat16),      # ONE patch per image, three stages in all: the short-pipeline waits (nst <= 3)
    (1, 24, 32, 32, 96, torch.bfloat16),     # Cout 96: two 80-cout blocks, the second 16 wide; 6 patches
    (5, 8, 32, 64, 64, torch.bfloat16),      # 10 patches, Cout 64
    (2, 16, 32, 64, 160, torch.bfloat16),    # 160-cout tile (three stage buffers): 8 patches, two cin tiles
    (1, 40, 48, 96, 320, torch.float16),     # two cout blocks, three cin tiles, fp16, 15 patches
    (4, 8, 16, 32, 136, torch.bfloat16),     # one patch per image, Cout not a multiple of 16
    (1, 80, 80, 160, 160, torch.bfloat16),   # the 80 x 80 C2f Bottleneck geometry (one image)
]


@pytest.mark.parametrize("case", WGRAD_PATCH_CASES)
def test_conv_wgrad_patch_form(case, sw):
    """The 8 x 16 patch form of the narrow tap-resident weight gradient (default where the map splits into whole patches) against autograd, against
    the linear 128-pixel-run form of the same kernel (CDET_WGRAD_PATCH=0) and against the im2col kernel: same bf16 products, fp32 sums in another
    order. Source and gradient are channel slices of wider buffers; the accumulate variant doubles the result."""
    ops = _ops()
    N, H, W, Ci, Co, dtype = case
    Cop = (Co + 7) // 8 * 8
    g = torch.Generator().manual_seed(9)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype).requires_grad_(True)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1)
    dy = _rt(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    xb = torch.randn(N, H, W, Ci + 24, generator=g).to(dtype).to(DEV)
    xb[..., 16:16 + Ci] = x.detach().permute(0, 2, 3, 1).to(dtype).to(DEV)
    xv = ops.View(xb, 16, Ci)
    yb = torch.randn(N, H, W, Cop + 8, generator=g).to(dtype).to(DEV)
    yb[..., 8:8 + Cop] = 0
    yb[..., 8:8 + Co] = dy.permute(0, 2, 3, 1).to(dtype).to(DEV)
    dyv = ops.View(yb, 8, Cop)
    scale = float(w.grad.abs().max())
    out = {}
    for name, env in (("patch", {}), ("linear", {"CDET_WGRAD_PATCH": "0"}), ("im2col", {"CDET_WGRAD_HALO": "0"})):
        for k_, v_ in env.items():
            sw(k_, int(v_))
        dw = torch.zeros(Co, Ci, 3, 3, device=DEV)
        ops.conv2d_wgrad(xv, dyv, dw, 3, 1)
        torch.cuda.synchronize()
        out[name] = dw
        for k_ in env:
            sw(k_, None)
    _close(out["patch"], w.grad, 2e-3, 2e-3 * scale)
    _close(out["patch"], out["linear"], 1e-4, 1e-4 * scale)
    _close(out["patch"], out["im2col"], 1e-4, 1e-4 * scale)
    again = out["patch"].clone()
    ops.conv2d_wgrad(xv, dyv, again, 3, 1, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(again, 2 * out["patch"])  # the same launch twice: identical sums, g + g exact


@pytest.mark.parametrize("case", [(2, 20, 20, 256, 160, torch.bfloat16), (1, 24, 40, 800, 320, torch.bfloat16), (3, 16, 16, 288, 136, torch.float16),
                                  (1, 13, 11, 1600, 200, torch.bfloat16), (1, 20, 20, 576, 640, torch.bfloat16)])
def test_conv_wgrad_1x1_transpose_read_kernel(case, sw):
    """1x1 weight gradient on wgrad_gemm_kernel (csrc/conv_wgrad_halo.hip) against autograd and against the im2col kernel; operands are
    channel slices of wider buffers; pixel counts that are not multiples of the 64-pixel stage; a cin tile that is partly empty."""
    ops = _ops()
    N, H, W, Ci, Co, dtype = case
    Cop = (Co + 7) // 8 * 8
    g = torch.Generator().manual_seed(17)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype).requires_grad_(False)
    w = _rt(torch.randn(Co, Ci, 1, 1, generator=g) / math.sqrt(Ci), dtype).requires_grad_(True)
    y = F.conv2d(x, w)
    dy = _rt(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    xb = torch.randn(N, H, W, Ci + 24, generator=g).to(dtype).to(DEV)
    xb[..., 16:16 + Ci] = x.permute(0, 2, 3, 1).to(dtype).to(DEV)
    xv = ops.View(xb, 16, Ci)
    yb = torch.randn(N, H, W, Cop + 8, generator=g).to(dtype).to(DEV)
    yb[..., 8:8 + Cop] = 0
    yb[..., 8:8 + Co] = dy.permute(0, 2, 3, 1).to(dtype).to(DEV)
    dyv = ops.View(yb, 8, Cop)
    sw("CDET_WGRAD_HALO", int("3"))  # the kernel for every Cout >= 128 (by default only Cout > 320 goes there)
    dw = torch.zeros(Co, Ci, 1, 1, device=DEV)
    ops.conv2d_wgrad(xv, dyv, dw, 1, 1)
    torch.cuda.synchronize()
    scale = float(w.grad.abs().max())
    _close(dw, w.grad, 2e-3, 2e-3 * scale)
    sw("CDET_WGRAD_HALO", int("0"))
    dw0 = torch.zeros_like(dw)
    ops.conv2d_wgrad(xv, dyv, dw0, 1, 1)
    torch.cuda.synchronize()
    sw("CDET_WGRAD_HALO", int("3"))
    _close(dw, dw0, 1e-4, 1e-4 * scale)
    ops.conv2d_wgrad(xv, dyv, dw, 1, 1, accumulate=True)
    torch.cuda.synchronize()
    _close(dw, 2 * w.grad, 2e-3, 4e-3 * scale)


@pytest.mark.parametrize("case", [(2, 20, 20, 64, 160, 5, torch.bfloat16), (1, 40, 40, 32, 136, 3, torch.float16), (1, 24, 24, 96, 320, 14, torch.bfloat16)])
def test_conv_wgrad_grouped_launch(case):
    """cdet_conv2d_wgrad_grouped: G layers of one geometry (x views at different channel offsets of wider buffers, like the Bottleneck
    inputs inside a C2f concat buffer) in one launch == G single launches, and == autograd; accumulate doubles."""
    ops = _ops()
    N, H, W, Ci, Co, G, dtype = case
    Cop = (Co + 7) // 8 * 8
    g = torch.Generator().manual_seed(11)
    wide = torch.randn(N, H, W, Ci * 3 + 8, generator=g).to(dtype).to(DEV)   # several layers read slices of this one
    items, refs = [], []
    for i in range(G):
        x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype)
        w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dtype).requires_grad_(True)
        y = F.conv2d(x, w, None, 1, 1)
        dy = _rt(torch.randn(y.shape, generator=g), dtype)
        y.backward(dy)
        if i % 2 == 0 and i < 6:
            off = 8 + (i // 2) * Ci
            wide[..., off:off + Ci] = x.permute(0, 2, 3, 1).to(dtype).to(DEV)
            xv = ops.View(wide, off, Ci)
        else:
            xv = ops.from_nchw(x.to(DEV), dtype)
        dyv = ops.new_act(N, H, W, Cop, dtype, zero=True)
        dyv.buf[..., :Co] = dy.permute(0, 2, 3, 1).to(dtype).to(DEV)
        items.append((xv, dyv, torch.zeros(Co, Ci, 3, 3, device=DEV)))
        refs.append(w.grad)
    ops.conv2d_wgrad_grouped(items, 3, 1)
    torch.cuda.synchronize()
    for (xv, dyv, dw), ref in zip(items, refs):
        _close(dw, ref, 2e-3, 2e-3 * float(ref.abs().max()))
        single = torch.zeros_like(dw)
        ops.conv2d_wgrad(xv, dyv, single, 3, 1)
        torch.cuda.synchronize()
        _close(dw, single, 1e-4, 1e-4 * float(ref.abs().max()))
    ops.conv2d_wgrad_grouped(items, 3, 1, accumulate=True)
    torch.cuda.synchronize()
    for (_, _, dw), ref in zip(items, refs):
        _close(dw, 2 * ref, 2e-3, 4e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("img_dtype", [torch.float32, torch.uint8])
def test_stem_conv_and_wgrad(img_dtype):
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, H, W, Co = 2, 36, 44, 80
    g = torch.Generator().manual_seed(3)
    img_u8 = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    img = img_u8.float() / 255
    w = (torch.randn(Co, 3, 3, 3, generator=g) / math.sqrt(27)).requires_grad_(True)
    y = F.conv2d(img, w, None, 2, 1)
    src = img_u8 if img_dtype == torch.uint8 else img
    # the MFMA stem kernel (csrc/stem_mfma.hip) rounds the scaled image and the weights to the 16-bit storage dtype like every other
    # convolution of the path; against a reference built from the SAME rounded operands only the fp32 summation order and the one
    # output rounding differ
    yq = F.conv2d(_rt(img, torch.bfloat16), _rt(w.detach(), torch.bfloat16), None, 2, 1)
    dst = ops.new_act(N, H // 2, W // 2, Co, torch.bfloat16)
    nblk = ops.stem_stat_blocks(N, H, W)
    stats = torch.zeros(nblk * 2 * Co, device=DEV)
    ops.stem_conv(src.to(DEV).contiguous(), w.detach().to(DEV), dst, stats=stats)
    torch.cuda.synchronize()
    _close(dst.nchw(), yq, 2 ** -7, 1e-3)
    _close(dst.nchw(), y.detach(), 2e-2, 2e-2)  # and close to the un-rounded convolution
    st = stats.view(nblk, 2, Co).sum(0).cpu()
    _close(st[0], yq.sum((0, 2, 3)), 1e-4, 1e-2)  # statistics come from the fp32 accumulators
    _close(st[1], (yq ** 2).sum((0, 2, 3)), 1e-4, 1e-2)
    dstb = ops.new_act(N, H // 2, W // 2, Co, torch.bfloat16)
    scale, bias = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    ops.stem_conv(src.to(DEV).contiguous(), w.detach().to(DEV), dstb, scale=scale.to(DEV), bias=bias.to(DEV), act=L.ACT_SILU)
    torch.cuda.synchronize()
    _close(dstb.nchw(), F.silu(yq * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)), 2 ** -7, 1e-2)
    # fp16 storage, half-precision image (the inference configuration)
    dsth = ops.new_act(N, H // 2, W // 2, Co, torch.float16)
    ops.stem_conv(img.half().to(DEV).contiguous(), w.detach().to(DEV), dsth)
    torch.cuda.synchronize()
    _close(dsth.nchw(), F.conv2d(_rt(img, torch.float16), _rt(w.detach(), torch.float16), None, 2, 1), 2 ** -10, 1e-3)
    dy = _rt(torch.randn(y.shape, generator=g), torch.bfloat16)
    y.backward(dy)
    dw = torch.zeros(Co, 3, 3, 3, device=DEV)
    ops.stem_conv_wgrad(src.to(DEV).contiguous(), ops.from_nchw(dy.to(DEV), torch.bfloat16), dw)
    torch.cuda.synchronize()
    # (the kernel multiplies the bf16-rounded image, like the forward: 2^-9 per element against this un-rounded reference; the test below
    # compares with the rounded operands at 1e-3)
    _close(dw, w.grad, 4e-3, 4e-3 * float(w.grad.abs().max()))


@pytest.mark.parametrize("case", [(2, 64, 64, 80, 160, torch.uint8, torch.bfloat16),      # YOLOv8x rows 0-1: 2.5 chunks, 160-cout tile
                                  (1, 96, 160, 16, 32, torch.float16, torch.float16),      # YOLOv8n rows 0-1, half image (inference form)
                                  (2, 72, 104, 48, 96, torch.float32, torch.bfloat16),     # partial tiles (18 x 26 outputs), 1.5 chunks
                                  (1, 128, 64, 96, 136, torch.bfloat16, torch.bfloat16),   # 3 full chunks, couts not a multiple of 32
                                  (3, 32, 32, 64, 160, torch.uint8, torch.float16)])
def test_fused_stem_and_second_row_equal_the_two_kernel_path(case):
    """csrc/stem_conv1.hip (eval plans: backbone rows 0 and 1 as one kernel, the stem's map kept in LDS) against cdet_stem_conv followed by
    cdet_conv2d_s2_tiled on the same operands: same arithmetic in the same summation order, so the outputs must carry the SAME BITS --
    incl. the second row's zero padding around the stem's map, partial tiles and a channel-slice destination. Also pinned against fp32
    F.conv2d of the rounded operands (one 16-bit rounding of the stem's map, one of the output)."""
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, H, W, c1, c2, img_dtype, dtype = case
    g = torch.Generator().manual_seed(61)
    img_u8 = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    img = img_u8 if img_dtype == torch.uint8 else (img_u8.float() / 255).to(img_dtype)
    w0 = torch.randn(c1, 3, 3, 3, generator=g) / math.sqrt(27)
    w1 = torch.randn(c2, c1, 3, 3, generator=g) / math.sqrt(9 * c1)
    s0, b0 = torch.rand(c1, generator=g) + 0.5, torch.randn(c1, generator=g) * 0.3
    s1, b1 = torch.rand(c2, generator=g) + 0.5, torch.randn(c2, generator=g) * 0.3
    imgd = img.to(DEV).contiguous()
    dev = lambda t: t.to(DEV)  # noqa: E731
    # two-kernel path
    mid = ops.new_act(N, H // 2, W // 2, c1, dtype)
    ops.stem_conv(imgd, dev(w0), mid, scale=dev(s0), bias=dev(b0), act=L.ACT_SILU)
    wf, _ = ops.pack_weight_tiled(dev(w1), dtype)
    two = ops.new_act(N, H // 4, W // 4, c2, dtype)
    assert ops.conv2d_s2_tiled_ok(mid, two)
    ops.conv2d_s2_tiled(mid, wf, two, scale=dev(s1), bias=dev(b1), act=L.ACT_SILU)
    # fused, into a channel slice of a wider buffer
    yb = torch.full((N, H // 4, W // 4, c2 + 24), 7.0, dtype=dtype, device=DEV)
    one = ops.View(yb, 16, c2)
    ops.stem_conv1(imgd, dev(w0), dev(w1), one, stem_scale=dev(s0), stem_bias=dev(b0), scale=dev(s1), bias=dev(b1))
    torch.cuda.synchronize()
    assert (yb[..., :16].float() == 7.0).all() and (yb[..., 16 + c2:].float() == 7.0).all(), "wrote outside its channel slice"
    diff = (one.torch().float() - two.torch().float()).abs()
    assert torch.equal(one.torch(), two.torch()), f"{int((diff > 0).sum())} of {diff.numel()} outputs differ, max {float(diff.max()):.3g}"
    # fp32 reference from the rounded operands
    x = img_u8.float() / 255 if img_dtype == torch.uint8 else img.float()
    m = _rt(F.silu(F.conv2d(_rt(x, dtype), _rt(w0, dtype), None, 2, 1) * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1)), dtype)
    ref = F.silu(F.conv2d(m, _rt(w1, dtype), None, 2, 1) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    _close(one.nchw(), ref, 2 ** -6, 2e-2)


@pytest.mark.parametrize("case", [(2, 36, 44, 80, torch.uint8, torch.bfloat16), (1, 20, 140, 80, torch.float32, torch.bfloat16),
                                  (3, 64, 64, 48, torch.uint8, torch.float16), (1, 18, 130, 64, torch.float16, torch.float16),
                                  (2, 22, 42, 80, torch.uint8, torch.bfloat16)])
def test_stem_weight_gradient_from_the_image(case):
    """csrc/stem_wgrad.hip against fp32 autograd on the same rounded operands (the kernel rounds the scaled image to the compute dtype like the
    forward does): ragged tiles in both directions, several tiles per row, a width that is not a multiple of 4 (byte-wise image decode), Cout below / at the 80 limit, dy rows wider than Cout, accumulate."""
    ops = _ops()
    N, H, W, Co, img_dtype, dtype = case
    g = torch.Generator().manual_seed(17)
    img_u8 = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    src = img_u8 if img_dtype == torch.uint8 else (img_u8.float() / 255).to(img_dtype)
    imgq = _rt(img_u8.float() * (1.0 / 255.0) if img_dtype == torch.uint8 else src.float(), dtype)
    dy = _rt(torch.randn(N, Co, H // 2, W // 2, generator=g), dtype)
    w = torch.zeros(Co, 3, 3, 3, requires_grad=True)
    F.conv2d(imgq, w, None, 2, 1).backward(dy)
    dyb = torch.full((N, H // 2, W // 2, Co + 8), float("nan"), dtype=dtype, device=DEV)  # channels beyond Cout are never read
    dyb[..., :Co] = dy.permute(0, 2, 3, 1).to(dtype).to(DEV)
    base = torch.randn(Co, 3, 3, 3, generator=g).to(DEV)
    got = ops.stem_conv_wgrad(src.to(DEV).contiguous(), ops.View(dyb, 0, Co), torch.empty(Co, 3, 3, 3, device=DEV))
    acc = ops.stem_conv_wgrad(src.to(DEV).contiguous(), ops.View(dyb, 0, Co), base.clone(), accumulate=True)
    torch.cuda.synchronize()
    sc = float(w.grad.abs().max())
    _close(got, w.grad, 0, 1e-3 * sc)
    _close(acc, base.cpu() + w.grad, 0, 1e-3 * sc)


@pytest.mark.parametrize("C,M_shape", [(80, (2, 12, 10)), (320, (3, 9, 7)), (640, (1, 5, 5))])
def test_bn_silu_forward_backward(C, M_shape):
    ops = _ops()
    N, H, W = M_shape
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(4)
    z32 = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    z = _rt(z32, dtype).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    res = _rt(torch.randn(N, C, H, W, generator=g), dtype)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.silu(F.batch_norm(z, rm_ref, rv_ref, gamma, beta, True, 0.03, 1e-3)) + res
    dy = _rt(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    # statistics come from the conv kernel in the product; emulate its partial-sum buffer (2 blocks)
    zf = z.detach().permute(0, 2, 3, 1).reshape(-1, C)
    half = zf.shape[0] // 2
    stats = torch.stack([torch.stack((zf[:half].sum(0), (zf[:half] ** 2).sum(0))), torch.stack((zf[half:].sum(0), (zf[half:] ** 2).sum(0)))]).to(DEV)
    mean, invstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    ops.bn_finalize(stats.contiguous(), 2, C, zf.shape[0], 1e-3, 0.03, rmd, rvd, mean, invstd)
    zv = ops.from_nchw(z.detach().to(DEV), dtype)
    yv = ops.new_act(N, H, W, C, dtype)
    ops.bn_silu_fwd(zv, mean, invstd, gamma.detach().to(DEV), beta.detach().to(DEV), yv, res=ops.from_nchw(res.to(DEV), dtype))
    torch.cuda.synchronize()
    _close(rmd, rm_ref, 1e-5, 1e-5)
    _close(rvd, rv_ref, 1e-4, 1e-5)
    _close(yv.nchw(), y.detach(), 2 ** -7, 2e-2)
    dz = ops.new_act(N, H, W, C, dtype)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_silu_bwd(ops.from_nchw(dy.to(DEV), dtype), zv, mean, invstd, gamma.detach().to(DEV), beta.detach().to(DEV), dz, dg, db)
    torch.cuda.synchronize()
    _close(dz.nchw(), z.grad, 2 ** -6, 2e-2 * float(z.grad.abs().max()))
    _close(dg, gamma.grad, 1e-3, 1e-3 * float(gamma.grad.abs().max()))
    _close(db, beta.grad, 1e-3, 1e-3 * float(beta.grad.abs().max()))


def test_upsample_copy_pool():
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 2, 6, 5, 32
    x = _rt(torch.randn(N, C, H, W, generator=g), dtype).requires_grad_(True)
    p4 = _rt(torch.randn(N, 16, 2 * H, 2 * W, generator=g), dtype)
    cat = torch.cat((F.interpolate(x, scale_factor=2, mode="nearest"), p4), 1)
    buf = ops.new_act(N, 2 * H, 2 * W, C + 16, dtype)
    ops.upsample2(ops.from_nchw(x.detach().to(DEV), dtype), buf.slice(0, C))
    ops.copy_channels(ops.from_nchw(p4.to(DEV), dtype), buf.slice(C, 16))
    torch.cuda.synchronize()
    assert torch.equal(buf.nchw().float().cpu(), cat.detach())
    dcat = _rt(torch.randn(cat.shape, generator=g), dtype)
    cat.backward(dcat)
    dbuf = ops.from_nchw(dcat.to(DEV), dtype)
    dx = ops.new_act(N, H, W, C, dtype)
    ops.upsample2_bwd(dbuf.slice(0, C), dx)
    torch.cuda.synchronize()
    _close(dx.nchw(), x.grad, 2 ** -7, 1e-2)


@pytest.mark.parametrize("hw", [(9, 11, 24), (20, 20, 24), (20, 20, 72), (19, 17, 24), (8, 8, 40), (40, 40, 24)])
def test_sppf_pool_chain_forward_backward(hw):
    """reference SPPF (models/common.py:230-245): three chained 5x5 max pools, their autograd routing to the first maximum.
    (20, 20) and (19, 17) take the one-tile form of the backward kernel, the others the 16x16 tiles. Round 4: maps up to 20 x 20 run the
    forward chain as ONE launch (sppf_pool3_kernel: plane in LDS, separable maxima; channel groups of 32 with a partial last group), the
    40 x 40 case three pool5 launches -- both must equal torch bit for bit."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    N = 2
    Hm, Wm, C2 = hw
    xs = _rt(torch.randn(N, C2, Hm, Wm, generator=g), dtype).requires_grad_(True)
    y1 = F.max_pool2d(xs, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    catp = torch.cat((xs, y1, y2, y3), 1)
    pb = ops.new_act(N, Hm, Wm, 4 * C2, dtype, zero=True)
    ops.copy_channels(ops.from_nchw(xs.detach().to(DEV), dtype), pb.slice(0, C2))
    ops.sppf_pool(pb, C2)
    torch.cuda.synchronize()
    assert torch.equal(pb.nchw().float().cpu(), catp.detach())
    dcp = _rt(torch.randn(catp.shape, generator=g), dtype)
    catp.backward(dcp)
    dpb = ops.from_nchw(dcp.to(DEV), dtype)
    ops.sppf_pool_bwd(pb, dpb, C2)
    torch.cuda.synchronize()
    # the three stages accumulate into the 16-bit gradient buffer: one rounding of the running sum (2^-9 of ITS size) per stage
    _close(dpb.slice(0, C2).nchw(), xs.grad, 2 ** -6, 2 ** -7 * float(xs.grad.abs().max()))


def test_detect_decode_matches_oracle():
    ops = _ops()
    from oracle import graph as og

    nc, N = 20, 2
    feats = [torch.from_numpy(f) for f in synth.synth_feats(31, N, 96, nc, "near")]
    want = og.detect_decode(feats, nc, (8.0, 16.0, 32.0))
    got = ops.detect_decode([f.permute(0, 2, 3, 1).contiguous().to(DEV) for f in feats], nc, (8.0, 16.0, 32.0))
    torch.cuda.synchronize()
    _close(got, want, 1e-4, 1e-4)


def _padded_gt(batch, bs, imgsz):
    from oracle import loss as ol

    t = ol.pad_targets(torch.from_numpy(batch["batch_idx"]), torch.from_numpy(batch["cls"]), torch.from_numpy(batch["prob"]),
                       torch.from_numpy(batch["bboxes"]), bs, torch.tensor([imgsz] * 4, dtype=torch.float32))
    if t.shape[1] == 0:
        t = torch.zeros(bs, 1, 6)
    return torch.cat((t[..., 0:1], t[..., 2:6]), -1).contiguous()


@pytest.mark.parametrize("name", list(synth.LOSS_CASES))
def test_det_loss_matches_reference_golden(name):
    """HIP loss vs the REAL reference's outputs (tests/golden/loss.npz): assignment bit-exact, loss/grad 1e-3 rel."""
    ops = _ops()
    arrays, meta = load_golden("loss")
    bs, imgsz, nc, npi, empty, seed, mode = synth.LOSS_CASES[name]
    batch = synth.make_batch(bs, max(npi, 1), nc, seed, empty if npi else tuple(range(bs)))
    feats = synth.synth_feats(seed, bs, imgsz, nc, mode)
    fd = [torch.from_numpy(f).permute(0, 2, 3, 1).contiguous().to(DEV) for f in feats]
    gt = _padded_gt(batch, bs, imgsz).to(DEV)
    loss5, dfe, asg = ops.det_loss(fd, gt, nc, meta[name]["gains"], (8.0, 16.0, 32.0), want_assign=True)
    torch.cuda.synchronize()
    p = f"{name}/"
    fg = arrays[p + "fg_mask"].astype(bool)
    assert np.array_equal(asg["fg_mask"].cpu().numpy().astype(bool), fg)
    assert np.array_equal(asg["target_gt_idx"].cpu().numpy(), arrays[p + "target_gt_idx"])
    if npi:
        assert np.array_equal(asg["target_labels"].cpu().numpy(), arrays[p + "target_labels"])
        assert np.allclose(asg["target_bboxes"].cpu().numpy(), arrays[p + "target_bboxes"], rtol=1e-6, atol=1e-4)
    assert np.abs(asg["target_scores"].cpu().numpy() - arrays[p + "target_scores"]).max() < 1e-4
    items = loss5.cpu().numpy()
    assert np.allclose(items[:4], arrays[p + "items"], rtol=1e-3, atol=1e-5), (items, arrays[p + "items"])
    assert abs(items[4] - float(arrays[p + "loss"])) <= 1e-3 * abs(float(arrays[p + "loss"])) + 1e-5
    for i in range(3):
        want = arrays[p + f"dfeat{i}"]
        got = dfe[i].permute(0, 3, 1, 2).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max() + 1e-6, (i, np.abs(got - want).max(), np.abs(want).max())


@pytest.mark.parametrize("name", list(synth.NMS_CASES) + ["ties"])
def test_nms_matches_oracle_bit_exact(name):
    ops = _ops()
    from oracle import nms as on

    _, meta = load_golden("nms")
    y = synth.ties_input() if name == "ties" else synth.nms_case_input(name)
    kw = dict(meta[name]["kw"])
    want = on.non_max_suppression(y, **kw)
    rows, cnt = ops.nms_batched(torch.from_numpy(y).to(DEV), **kw)
    torch.cuda.synchronize()
    cnt = cnt.cpu().numpy()
    assert cnt.tolist() == [w.shape[0] for w in want] == meta[name]["counts"]
    for i, w in enumerate(want):
        assert np.array_equal(rows[i, :cnt[i]].cpu().numpy(), w), (name, i)


def test_sgd_ema_step_matches_oracle():
    import ctypes as C

    from cerberusdet_amd import _lib as L
    from oracle import optim as oo

    lib = L.load()
    g = torch.Generator().manual_seed(9)
    keys = ["blocks.0.model.0.conv.weight", "blocks.0.model.0.bn.weight", "blocks.0.model.0.bn.bias", "blocks.3.cv1.conv.weight"]
    shapes = [(16, 3, 3, 3), (16,), (16,), (64, 32, 1, 1)]
    w = {k: torch.randn(s, generator=g) for k, s in zip(keys, shapes)}
    gr = {k: torch.randn(s, generator=g) * 3 for k, s in zip(keys, shapes)}
    ema = {k: v.clone() + 0.1 for k, v in w.items()}
    serving = {0: 2, 3: 1}
    lrs = (0.01, 0.02, 0.03)
    wd = {k: w[k].clone().to(DEV) for k in keys}
    gd = {k: gr[k].clone().to(DEV) for k in keys}
    md = {k: torch.zeros_like(wd[k]) for k in keys}
    ed = {k: ema[k].clone().to(DEV) for k in keys}
    mom_ref, upd = {}, 0
    wref, eref = {k: v.clone() for k, v in w.items()}, {k: v.clone() for k, v in ema.items()}
    for step in range(2):
        slots = (L.ParamSlot * len(keys))()
        for i, k in enumerate(keys):
            grp = oo.param_group(k)
            slots[i].p, slots[i].g, slots[i].mom, slots[i].ema = wd[k].data_ptr(), gd[k].data_ptr(), md[k].data_ptr(), ed[k].data_ptr()
            slots[i].n, slots[i].group = wd[k].numel(), grp
            slots[i].weight_decay = 0.00037 if grp == 0 else 0.0
            slots[i].inv_div, slots[i].first_step = 1.0 / serving[oo.block_of(k)], int(step == 0)
        sdev = torch.frombuffer(bytearray(bytes(slots)), dtype=torch.uint8).to(DEV)
        out = torch.zeros(1 + 32 * len(keys), device=DEV)
        L.check(lib.cdet_grad_sqnorm(sdev.data_ptr(), len(keys), out.data_ptr(), None, torch.cuda.current_stream().cuda_stream))
        d = oo.ema_decay(upd + 1)
        L.check(lib.cdet_sgd_ema_step(sdev.data_ptr(), len(keys), out.data_ptr(), 10.0, (C.c_float * 3)(*lrs), 3, 0.952, d, None, None,
                                      torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        total = oo.optimizer_step(wref, {k: v.clone() for k, v in gr.items()}, mom_ref, serving, lr=lrs, momentum=0.952, weight_decay=0.00037)
        upd = oo.ema_update(eref, wref, upd)
        assert abs(math.sqrt(float(out[0])) - total) < 1e-3 * total
        for k in keys:
            _close(wd[k], wref[k], 1e-5, 1e-6)
            _close(ed[k], eref[k], 1e-5, 1e-6)
            assert float(gd[k].abs().max()) == 0.0
            gd[k].copy_(gr[k])


@pytest.mark.parametrize("scaled", [False, True])
def test_sgd_ema_step_skips_on_a_non_finite_gradient_like_gradscaler(scaled):
    """reference trainers/averaging.py:61, 205-223 (amp.GradScaler: unscale_ -> clip -> step -> update -> zero_grad -> ema.update): a non-finite
    gradient ANYWHERE cancels the optimizer step -- every weight and momentum buffer keeps its bits, the gradients are zeroed, the EMA lerp still
    runs -- the scale backs off, and the next step is an ordinary one. Six steps with inf / NaN injected into different tensors at steps 1, 2 and 4,
    against oracle/optim.py (whose scaler restatement is pinned to torch's own GradScaler on the CPU, tests/test_oracle_golden.py). `scaled`: the
    fp16 plans' loss scaling (gradients arrive multiplied by the scale; growth every 2 good steps here) / the bf16 plans' fixed scale 1."""
    import ctypes as C

    from cerberusdet_amd import _lib as L
    from oracle import optim as oo

    lib = L.load()
    g = torch.Generator().manual_seed(19)
    keys = ["blocks.0.model.0.conv.weight", "blocks.0.model.0.bn.weight", "blocks.0.model.0.bn.bias", "blocks.3.cv1.conv.weight"]
    shapes = [(16, 3, 3, 3), (16,), (16,), (64, 32, 3, 3)]
    w = {k: torch.randn(s, generator=g) for k, s in zip(keys, shapes)}
    ema = {k: v.clone() + 0.1 for k, v in w.items()}
    serving = {0: 2, 3: 1}
    lrs = (0.01, 0.02, 0.03)
    wd = {k: w[k].clone().to(DEV) for k in keys}
    gd = {k: torch.zeros_like(wd[k]) for k in keys}
    md = {k: torch.zeros_like(wd[k]) for k in keys}
    ed = {k: ema[k].clone().to(DEV) for k in keys}
    interval = 2 if scaled else 0
    ref_sc = oo.GradScalerState(scale=65536.0 if scaled else 1.0, growth_interval=interval)
    sc_dev = torch.tensor([ref_sc.scale, 0.0, 0.0, 0.0], device=DEV)
    mom_ref, upd = {}, 0
    wref, eref = {k: v.clone() for k, v in w.items()}, {k: v.clone() for k, v in ema.items()}
    poison = {1: (keys[3], float("inf")), 2: (keys[1], float("nan")), 4: (keys[0], float("-inf"))}
    st = torch.cuda.current_stream().cuda_stream
    for step in range(6):
        gr = {k: torch.randn(s, generator=g) * 3 * ref_sc.scale for k, s in zip(keys, shapes)}  # scaled gradients, as the backward leaves them
        if step in poison:
            k_, v_ = poison[step]
            gr[k_].view(-1)[7] = v_
        for k in keys:
            gd[k].copy_(gr[k])
        slots = (L.ParamSlot * len(keys))()
        for i, k in enumerate(keys):
            grp = oo.param_group(k)
            slots[i].p, slots[i].g, slots[i].mom, slots[i].ema = wd[k].data_ptr(), gd[k].data_ptr(), md[k].data_ptr(), ed[k].data_ptr()
            slots[i].n, slots[i].group = wd[k].numel(), grp
            slots[i].weight_decay = 0.00037 if grp == 0 else 0.0
            slots[i].inv_div, slots[i].first_step = 1.0 / serving[oo.block_of(k)], int(step == 0)
        sdev = torch.frombuffer(bytearray(bytes(slots)), dtype=torch.uint8).to(DEV)
        out = torch.zeros(1 + 32 * len(keys), device=DEV)
        before_w = {k: wd[k].clone() for k in keys}
        before_m = {k: md[k].clone() for k in keys}
        scp = sc_dev.data_ptr() if scaled else None
        L.check(lib.cdet_grad_sqnorm(sdev.data_ptr(), len(keys), out.data_ptr(), scp, st))
        d = oo.ema_decay(upd + 1)
        # (scaled: scaler.update() is a launch of its own behind the update; unscaled: the skip counter / found_inf words ride on the update launch)
        L.check(lib.cdet_sgd_ema_step(sdev.data_ptr(), len(keys), out.data_ptr(), 10.0, (C.c_float * 3)(*lrs), 3, 0.952, d, scp,
                                      None if scaled else sc_dev.data_ptr() + 8, st))
        if scaled:
            L.check(lib.cdet_scaler_update(sc_dev.data_ptr(), out.data_ptr(), 2.0, 0.5, interval, st))
        torch.cuda.synchronize()
        total = oo.optimizer_step(wref, {k: v.clone() for k, v in gr.items()}, mom_ref, serving, lr=lrs, momentum=0.952, weight_decay=0.00037,
                                  scaler=ref_sc)
        upd = oo.ema_update(eref, wref, upd)
        for k in keys:
            assert float(gd[k].abs().max()) == 0.0, "zero_grad() happens on a skipped step too"
            if step in poison:
                assert torch.equal(wd[k], before_w[k]) and torch.equal(md[k], before_m[k]), f"step {step}: {k} moved on a skipped step"
            _close(wd[k], wref[k], 1e-5, 1e-6)
            _close(ed[k], eref[k], 1e-5, 1e-6)
            if k in mom_ref:
                _close(md[k], mom_ref[k], 1e-5, 1e-6)
        if step in poison:
            assert not math.isfinite(float(out[0])) and not math.isfinite(total)
        else:
            assert abs(math.sqrt(float(out[0])) - total) < 1e-3 * total
        scv = sc_dev.tolist()
        assert scv[0] == ref_sc.scale and int(scv[1]) == ref_sc.growth_tracker and int(scv[2]) == ref_sc.skipped and bool(scv[3]) == (step in poison), (step, scv)
    assert ref_sc.skipped == 3 and all(bool(torch.isfinite(wd[k]).all()) and bool(torch.isfinite(ed[k]).all()) for k in keys)


def test_pack_weights_batched_equals_per_item_pack():
    """cdet_pack_weights_batched (one launch, LDS tile transpose, valid elements only) against cdet_pack_weight per operand."""
    import ctypes as C

    ops = _ops()
    from cerberusdet_amd import _lib as L

    lib = L.load()
    g = torch.Generator().manual_seed(11)
    shapes = [(80, 8, 3, 3, 80), (160, 80, 3, 3, 160), (320, 400, 1, 1, 320), (20, 320, 1, 1, 24), (64, 80, 1, 1, 64), (48, 72, 3, 3, 48)]
    for dtype in (torch.bfloat16, torch.float16):
        ws, refs, outs = [], [], []
        arr = (L.PackItem * len(shapes))()
        blk = 0
        for it, (O, I, kh, kw, Op) in zip(arr, shapes):
            w = torch.randn(O, I, kh, kw, generator=g).to(DEV)
            ws.append(w)
            refs.append((ops.pack_weight(w, dtype, o_pad=Op), ops.pack_weight(w, dtype, transpose=True, o_pad=Op)))
            wf, wt = torch.zeros_like(refs[-1][0]), torch.zeros_like(refs[-1][1])
            outs.append((wf, wt))
            nb = ((O + 31) // 32) * ((I + 31) // 32)
            it.w_oihw, it.w_fwd, it.w_dgrad = w.data_ptr(), wf.data_ptr(), wt.data_ptr()
            it.O, it.O_pad, it.I, it.kh, it.kw, it.first_block, it.n_blocks = O, Op, I, kh, kw, blk, nb
            blk += nb
        dtab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
        L.check(lib.cdet_pack_weights_batched(dtab.data_ptr(), len(shapes), blk, ops.dt(dtype), ops.stream()), "cdet_pack_weights_batched")
        torch.cuda.synchronize()
        for (rf, rt), (wf, wt), shp in zip(refs, outs, shapes):
            assert torch.equal(rf.view(torch.int16), wf.view(torch.int16)), ("fwd", shp, dtype)
            assert torch.equal(rt.view(torch.int16), wt.view(torch.int16)), ("dgrad", shp, dtype)


@pytest.mark.parametrize("with_scale", [False, True])
def test_merge_tasks_matches_oracle_bit_exact(with_scale):
    """cdet_merge_tasks (class remap + cross-task suppression + scale_boxes().round()) against the CPU restatement of
    utils/general.py:484-554 / 313-357 on boxes with many cross-task overlaps, score ties and an all-deleted image."""
    from oracle import nms as onms

    ops = _ops()
    rng = np.random.default_rng(5)
    N, T, max_det, ncs = 5, 3, 40, [4, 3, 5]
    offs = [0, 4, 7]
    cmap = {f"t{t}": {i: i + offs[t] for i in range(ncs[t])} for t in range(T)}
    rows = [np.zeros((N, max_det, 6), np.float32) for _ in range(T)]
    cnts = [np.zeros(N, np.int32) for _ in range(T)]
    for n in range(N):
        base = rng.uniform(0, 500, (max_det, 2)).astype(np.float32)
        wh = rng.uniform(20, 120, (max_det, 2)).astype(np.float32)
        for t in range(T):
            k = int(rng.integers(max_det // 2, max_det + 1)) if n != 3 else 0 if t == 1 else 6
            jit = rng.uniform(-3, 3, (max_det, 4)).astype(np.float32) * (t > 0)
            b = np.concatenate([base, base + wh], 1) + jit
            if n == 4:  # exact duplicates with tied scores across all tasks
                b = np.concatenate([base, base + wh], 1)
            sc = rng.uniform(0.3, 0.99, max_det).astype(np.float32)
            if n == 4:
                sc = np.linspace(0.9, 0.5, max_det).astype(np.float32)
            order = np.argsort(-sc, kind="stable")
            rows[t][n, :k, :4] = b[order][:k]
            rows[t][n, :k, 4] = sc[order][:k]
            rows[t][n, :k, 5] = rng.integers(0, ncs[t], k)
            cnts[t][n] = k
    shapes = [(480, 640), (720, 1280), (333, 500), (640, 640), (1080, 1920)]
    scale = None
    if with_scale:
        sc_rows = []
        for shp in shapes:
            gain = min(640 / shp[0], 640 / shp[1])
            sc_rows.append([gain, (640 - shp[1] * gain) / 2, (640 - shp[0] * gain) / 2, shp[0], shp[1]])
        scale = torch.tensor(sc_rows, dtype=torch.float32, device=DEV)
    out, cnt = ops.merge_tasks([torch.from_numpy(r).to(DEV) for r in rows], [torch.from_numpy(c).to(DEV) for c in cnts], offs, 0.8, scale)
    torch.cuda.synchronize()
    out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
    n_deleted = 0
    for n in range(N):
        det = np.concatenate([np.concatenate([rows[t][n, :cnts[t][n], :5], rows[t][n, :cnts[t][n], 5:6] + offs[t]], 1) for t in range(T)], 0)
        want = onms.nms_between_tasks(det.copy(), cmap, 0.8) if len(det) else det
        n_deleted += len(det) - len(want)
        if with_scale and len(want):
            want = want.copy()
            want[:, :4] = np.round(onms.scale_boxes((640, 640), want[:, :4], shapes[n]))
        assert cnt[n] == len(want), (n, cnt[n], len(want))
        assert np.array_equal(out[n, :cnt[n]], want.astype(np.float32)), n
    assert n_deleted > 50  # the suppression really fired


def test_nms_between_tasks_api_matches_reference_golden():
    """The reference-signature wrapper (rows [n,6] with global class ids, single image) on the reference's own golden case."""
    from cerberusdet_amd.utils.general import nms_between_tasks
    from oracle import nms as on
    from util import GOLDEN

    arrays = dict(np.load(GOLDEN / "nms.npz"))
    _, _, names, _ = synth.predict_inputs()
    cmap, _ = on.categories_map(names)
    out = nms_between_tasks(torch.from_numpy(arrays["between/in"]), cmap, 0.8)
    assert not out.is_cuda and np.array_equal(out.numpy(), arrays["between/out"])
    out = nms_between_tasks(torch.from_numpy(arrays["between/in"]).to(DEV), cmap, 0.8)
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), arrays["between/out"])


def test_match_predictions_and_ap_match_reference_golden():
    """cdet_match_predictions (whole batch, one launch) against val.process_batch of the REAL reference (tests/golden/val.npz), the
    reference-signature wrapper, and the host AP against utils/metrics.ap_per_class."""
    from cerberusdet_amd.utils.metrics import ap_per_class, process_batch
    from util import GOLDEN

    ops = _ops()
    g = dict(np.load(GOLDEN / "val.npz"))
    cases = [synth.val_case(*c) for c in synth.VAL_CASES]
    N, max_det = len(cases), max(max(c[0].shape[0] for c in cases), 1)
    rows = np.zeros((N, max_det, 6), np.float32)
    cnt = np.zeros(N, np.int32)
    start = np.zeros(N + 1, np.int32)
    for i, (det, lab) in enumerate(cases):
        rows[i, :len(det)], cnt[i], start[i + 1] = det, len(det), start[i] + len(lab)
    labels = np.concatenate([lab for _, lab in cases], 0)
    iouv = torch.from_numpy(g["iouv"]).to(DEV)
    correct = ops.match_predictions(torch.from_numpy(rows).to(DEV), torch.from_numpy(cnt).to(DEV), torch.from_numpy(labels).to(DEV),
                                    torch.from_numpy(start).to(DEV), iouv, max_labels=max(len(lab) for _, lab in cases))
    torch.cuda.synchronize()
    correct = correct.cpu().numpy()
    stats = []
    for i, (det, lab) in enumerate(cases):
        want = g[f"case{i}/correct"]
        assert np.array_equal(correct[i, :len(det)].astype(bool), want), i
        assert not correct[i, len(det):].any()
        if len(det):
            single = process_batch(torch.from_numpy(det), torch.from_numpy(lab), iouv.cpu())
            assert np.array_equal(single.numpy(), want), i
        stats.append((want, det[:, 4], det[:, 5], lab[:, 0]))
    tp, conf, pcls, tcls = [np.concatenate(x, 0) for x in zip(*stats)]
    r = ap_per_class(tp, conf, pcls, tcls)
    for k, v in zip(("tp", "fp", "p", "r", "f1", "ap", "classes"), r):
        assert np.allclose(np.asarray(v, np.float64), g[f"ap/{k}"].astype(np.float64), rtol=1e-9, atol=1e-12), k


def test_val_run_smoke_on_tiny_model():
    """val.run end to end on the tiny golden model: finite metrics, every image seen, labels counted."""
    import copy

    from cerberusdet_amd import val
    from cerberusdet_amd.models import CerberusDet
    from util import load_golden

    _, mmeta = load_golden("model_tiny2")
    m = CerberusDet(mmeta["tasks"], mmeta["nc"], cfg=copy.deepcopy(mmeta["cfg"]), verbose=False)
    m.sequential_split(mmeta["cfg"]["cerber"], "cpu")
    sd = m.state_dict()
    m.load_state_dict({k: torch.from_numpy(synth.det_tensor(mmeta["seed"], k, v.shape)) for k, v in sd.items()})
    m = m.to(DEV).eval()
    batches = []
    for i in range(2):
        b = synth.make_batch(4, 3, mmeta["nc"][0], 700 + i)
        batches.append(dict(img=torch.from_numpy(synth.det_image(800 + i, 4, 128)), **{k: torch.from_numpy(v) for k, v in b.items()}))
    res = val.run(m, mmeta["tasks"][0], batches, half=False)
    assert res["seen"] == 8 and int(res["nt"].sum()) == 24
    assert all(np.isfinite(res[k]) and 0.0 <= res[k] <= 1.0 for k in ("mp", "mr", "map50", "map"))


def test_detect_decode_four_lane_form_carries_the_bits_of_the_one_thread_form():
    """cdet_detect_decode picks the four-lanes-per-anchor kernel (round 5) for fp32 maps whose row length is a multiple of 4 and the one-thread-per-anchor
    kernel otherwise: the same head maps in an 88-float and in an 85-float row layout (nc = 21) must give the same bits -- per side the DFL softmax
    expectation runs in the same order in both forms. Batch 3, levels 20 x 12 / 10 x 6 / 5 x 3 (a last partial block)."""
    from cerberusdet_amd import ops

    nc, N = 21, 3
    g = torch.Generator().manual_seed(12)
    shapes = [(20, 12), (10, 6), (5, 3)]
    maps = [torch.randn(N, h, w, 64 + nc, generator=g) * 3 for h, w in shapes]
    outs = []
    for ld in (88, 85):
        feats = []
        for m in maps:
            f = torch.full((N, m.shape[1], m.shape[2], ld), float("nan"))
            f[..., :64 + nc] = m
            feats.append(f.to(DEV))
        outs.append(ops.detect_decode(feats, nc, (8.0, 16.0, 32.0)))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
