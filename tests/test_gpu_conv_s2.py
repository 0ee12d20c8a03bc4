"""Stride-2 3x3 convolutions on the parity-plane kernels (csrc/conv_vt.hip: cdet_conv2d_s2_tiled, cdet_conv2d_s2_tiled_dgrad) against
an fp32 reference of the same op (F.conv2d and its autograd on the CPU; models/common.py:57-62 with s = 2).

Inputs are exactly representable in the storage dtype: the differences are the fp32 accumulation order and one output rounding (2^-7
relative for 16-bit outputs). Shapes cover both tile forms (256 consecutive output pixels for narrow maps, 16 x 16 patches), tiles
that straddle rows and images, M not a multiple of 256, 96- and 160-row weight blocks, Cout not a multiple of the block, a partial last
32-channel chunk, channel-slice source / destination, the fused epilogue (scale / bias / SiLU / residual), BN partial sums, and for
the data gradient the four parity classes incl. the gradient fan-in (accumulate) form."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rt(x, dtype):
    return x.to(dtype).float()


def _close(a, b, rtol, atol):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), f"max err {err.max():.4g} (ref max {b.abs().max():.4g}), {int(bad.sum())}/{bad.numel()} out of tolerance"


CASES = [
    # N, H, W (input), Cin, Cout, dtype
    (2, 40, 40, 64, 160, torch.bfloat16),    # output 20 x 20: linear tiles, 3.1 tiles, straddle images
    (1, 80, 80, 32, 320, torch.bfloat16),    # output 40 x 40 (the widest linear form), two cout blocks
    (3, 14, 10, 96, 48, torch.bfloat16),     # tiny odd output 7 x 5: several images per tile, 96-row weight block
    (1, 32, 64, 64, 160, torch.bfloat16),    # output 16 x 32: PATCH tiles
    (2, 64, 32, 80, 80, torch.bfloat16),     # patch + 3 cout fragments + partial last chunk (Cin = 80)
    (1, 26, 42, 160, 200, torch.float16),    # output 13 x 21, fp16, Cout not a multiple of 160
    (1, 160, 160, 32, 96, torch.float16),    # output 80 x 80: patch, fp16
    (2, 12, 12, 416, 64, torch.bfloat16),    # 13 chunks, output 6 x 6
]


@pytest.mark.parametrize("case", CASES)
def test_s2_forward_raw_stats_and_fused_epilogue(case):
    from cerberusdet_amd import _lib as L
    from cerberusdet_amd import ops

    N, H, W, Ci, Co, dtype = case
    g = torch.Generator().manual_seed(21)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dtype)
    ref_raw = F.conv2d(x, w, None, 2, 1)
    Ho, Wo = ref_raw.shape[2:]
    xb = torch.full((N, H, W, Ci + 16), float("nan"), dtype=dtype, device=DEV)  # the slice's neighbours are NaN: never read
    xb[..., 8:8 + Ci] = x.permute(0, 2, 3, 1).to(dtype).to(DEV)
    src = ops.View(xb, 8, Ci)
    wf, _ = ops.pack_weight_tiled(w.to(DEV), dtype)
    dst = ops.new_act(N, Ho, Wo, Co, dtype)
    assert ops.conv2d_s2_tiled_ok(src, dst)
    nblk = ops.conv_s2_tiled_stat_blocks(src, dst)
    stats = torch.zeros(nblk * 2 * Co, device=DEV)
    ops.conv2d_s2_tiled(src, wf, dst, stats=stats)
    torch.cuda.synchronize()
    _close(dst.nchw(), ref_raw, 2 ** -7, 1e-3)
    st = stats.view(nblk, 2, Co).sum(0).cpu()
    _close(st[0], ref_raw.sum((0, 2, 3)), 1e-3, 1e-2)
    _close(st[1], (ref_raw ** 2).sum((0, 2, 3)), 1e-3, 1e-2)
    # fused epilogue into a channel slice
    scale = torch.rand(Co, generator=g) + 0.5
    bias = torch.randn(Co, generator=g) * 0.1
    res = _rt(torch.randn(N, Co, Ho, Wo, generator=g), dtype)
    yb = torch.full((N, Ho, Wo, Co + 16), 7.0, dtype=dtype, device=DEV)
    dsl = ops.View(yb, 8, Co)
    ops.conv2d_s2_tiled(src, wf, dsl, scale=scale.to(DEV), bias=bias.to(DEV), act=L.ACT_SILU, res=ops.from_nchw(res.to(DEV), dtype))
    torch.cuda.synchronize()
    ref = F.silu(ref_raw * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)) + res
    _close(dsl.nchw(), ref, 2 ** -7, 2e-2)
    assert (yb[..., :8].float() == 7.0).all() and (yb[..., 8 + Co:].float() == 7.0).all(), "conv wrote outside its channel slice"
    # the generic kernel on the same operands agrees to one rounding
    alt = ops.new_act(N, Ho, Wo, Co, dtype)
    ops.conv2d(src, ops.pack_weight(w.to(DEV), dtype), alt, 3, 2)
    torch.cuda.synchronize()
    _close(dst.torch(), alt.torch(), 2 ** -6, 2e-3)


@pytest.mark.parametrize("case", CASES)
def test_s2_data_gradient_four_parity_classes(case):
    from cerberusdet_amd import _lib as L
    from cerberusdet_amd import ops

    N, H, W, Ci, Co, dtype = case
    g = torch.Generator().manual_seed(22)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype).requires_grad_(True)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dtype)
    y = F.conv2d(x, w, None, 2, 1)
    dy = _rt(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    _, wd = ops.pack_weight_tiled(w.to(DEV), dtype, fwd=False, dgrad=True)
    dyv = ops.from_nchw(dy.to(DEV), dtype)
    dxb = torch.full((N, H, W, Ci + 24), 5.0, dtype=dtype, device=DEV)
    dx = ops.View(dxb, 16, Ci)
    assert ops.conv2d_s2_tiled_ok(dyv, dx, L.CONV_DGRAD)
    ops.conv2d_s2_tiled_dgrad(dyv, wd, dx)
    torch.cuda.synchronize()
    tol = float(x.grad.abs().max())
    _close(dx.nchw(), x.grad, 2 ** -7, 2e-3 * tol)
    assert (dxb[..., :16].float() == 5.0).all() and (dxb[..., 16 + Ci:].float() == 5.0).all()
    # fan-in: the gradient already there is added (every pixel of dX belongs to exactly one class)
    prev = ops.from_nchw(_rt(torch.randn(N, Ci, H, W, generator=g), dtype).to(DEV), dtype)
    out = ops.new_act(N, H, W, Ci, dtype)
    out.buf.copy_(prev.buf)
    ops.conv2d_s2_tiled_dgrad(dyv, wd, out, res=out)
    torch.cuda.synchronize()
    _close(out.nchw(), x.grad + prev.nchw().float().cpu(), 2 ** -7, 4e-3 * tol)
    # the generic parity-class kernel agrees
    alt = ops.new_act(N, H, W, Ci, dtype)
    ops.conv2d(dyv, ops.pack_weight(w.to(DEV), dtype, transpose=True), alt, 3, 2, mode=L.CONV_DGRAD)
    torch.cuda.synchronize()
    _close(dx.torch(), alt.torch(), 2 ** -6, 2e-3 * tol)


WGRAD_CASES = [
    # N, H, W (input), Cin, Cout, dtype: all taken by csrc/conv_wgrad_s2.hip (patches of 8 x 16 outputs at least 75 % full)
    (2, 32, 64, 64, 160, torch.bfloat16),     # output 16 x 32: full patches, two cin tiles, one cout block
    (1, 80, 80, 80, 320, torch.bfloat16),     # output 40 x 40: ragged patch columns (40 = 2.5 x 16), half-empty last cin tile, 2 cout blocks
    (3, 48, 32, 32, 200, torch.bfloat16),     # output 24 x 16, Cout not a multiple of 160
    (2, 28, 60, 96, 128, torch.float16),      # output 14 x 30: ragged rows and columns, fp16
    (1, 160, 160, 32, 160, torch.bfloat16),   # output 80 x 80: many patches per split
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_s2_weight_gradient_parity_planes(case, sw):
    """dW of the stride-2 3x3 convolution (autograd's convolution_backward(weight) of models/common.py:57 with s = 2) against fp32 autograd
    on the same 16-bit-rounded operands (1e-3 of the tensor scale), against the round-1 im2col kernel (1e-4), and the accumulate form."""
    from cerberusdet_amd import _lib as L
    from cerberusdet_amd import ops
    import ctypes as C

    N, H, W, Ci, Co, dtype = case
    g = torch.Generator().manual_seed(33)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype)
    Ho, Wo = H // 2, W // 2
    dy = _rt(torch.randn(N, Co, Ho, Wo, generator=g), dtype)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, 2, 1).backward(dy)
    want = w.grad
    xb = torch.full((N, H, W, Ci + 16), float("nan"), dtype=dtype, device=DEV)  # the slice's neighbours are NaN: never read
    xb[..., 8:8 + Ci] = x.permute(0, 2, 3, 1).to(dtype).to(DEV)
    src = ops.View(xb, 8, Ci)
    Cp = (Co + 7) // 8 * 8
    dyb = torch.zeros((N, Ho, Wo, Cp + 8), dtype=dtype, device=DEV)
    dyb[..., 8:8 + Co] = dy.permute(0, 2, 3, 1).to(dtype).to(DEV)
    dyv = ops.View(dyb, 8, Cp)
    d = ops.conv_desc(src, dyv, 3, 2)
    d.Cd = Co
    lib = L.load()
    sw("CDET_WGRAD_S2", int("0"))
    n_old = lib.cdet_conv2d_wgrad_ws_elems(C.byref(d))
    old = ops.conv2d_wgrad(src, dyv, torch.empty(Co, Ci, 3, 3, device=DEV), 3, 2)
    sw("CDET_WGRAD_S2", None)
    n_new = lib.cdet_conv2d_wgrad_ws_elems(C.byref(d))
    assert n_new != n_old or Co * Ci < 160 * 32, "the case must be taken by the parity-plane kernel (different slab plan)"
    base = torch.randn(Co, Ci, 3, 3, generator=g).to(DEV)
    got = ops.conv2d_wgrad(src, dyv, torch.empty(Co, Ci, 3, 3, device=DEV), 3, 2)
    acc = ops.conv2d_wgrad(src, dyv, base.clone(), 3, 2, accumulate=True)
    torch.cuda.synchronize()
    sc = float(want.abs().max())
    _close(got, want, 0, 1e-3 * sc)
    _close(got, old, 0, 1e-4 * sc)
    _close(acc, base.cpu() + want, 0, 1e-3 * sc)
