"""Tap-resident convolution (csrc/conv_halo.hip, cdet_conv2d_tiled) against an fp32 CPU reference of the same op.

Inputs are exactly representable in the storage dtype, so the differences are the fp32 accumulation order and the one
output rounding: 2^-7 relative for 16-bit outputs. Shapes cover: tiles that straddle image rows and images, M not a
multiple of 256, W from 5 to 80 (2- and 3-stage weight rings), Cout not a multiple of 160, 1x1, concat-slice
source / destination, residual, BN partial sums, and the data gradient through the flipped/transposed operand.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from cerberusdet_amd import ops

    return ops


def _rt(x, dtype):
    return x.to(dtype).float()


def _close(a, b, rtol, atol):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), f"max err {err.max():.4g} (ref max {b.abs().max():.4g}), {int(bad.sum())}/{bad.numel()} out of tolerance"


CASES = [
    # N, H, W, Cin, Cout, k, dtype
    (2, 20, 20, 320, 320, 3, torch.bfloat16),   # two cout blocks, 10 chunks, tiles straddle images (800 px = 3.1 tiles)
    (1, 40, 40, 64, 160, 3, torch.bfloat16),    # W = 40: 3-stage ring, 6.25 tiles
    (1, 24, 80, 32, 160, 3, torch.bfloat16),    # W = 80: 2-stage ring, one chunk
    (3, 7, 5, 96, 48, 3, torch.bfloat16),       # tiny odd map: every tile holds several images; Cout < 160 (masked rows)
    (2, 16, 16, 320, 160, 1, torch.bfloat16),   # 1x1
    (1, 12, 12, 416, 320, 1, torch.float16),    # fp16, 13 chunks
    (1, 13, 21, 160, 200, 3, torch.float16),    # Cout not a multiple of 160 (second cout block mostly padding)
    (1, 33, 95, 32, 32, 3, torch.bfloat16),     # widest supported LINEAR map for 3x3 (7 X pieces per wave); 96-row weight blocks
    (1, 48, 64, 64, 160, 3, torch.bfloat16),    # PATCH mode (16 x 16 pixel tiles, halo 18 x 18): 12 patches
    (2, 32, 80, 96, 80, 3, torch.bfloat16),     # patch mode + 3 cout fragments (80 channels in a 96-row block)
    (1, 160, 160, 32, 96, 3, torch.float16),    # 160-wide map: patch mode only (linear halo would not fit), fp16
    (1, 40, 40, 64, 80, 3, torch.bfloat16),     # linear mode + 3 cout fragments
    (2, 16, 16, 96, 80, 1, torch.bfloat16),     # 1x1 + 3 cout fragments
    (1, 32, 48, 80, 80, 3, torch.bfloat16),     # Cin = 80: last 32-channel chunk half empty (zero-filled lanes), patch mode
    (2, 20, 20, 80, 80, 3, torch.bfloat16),     # Cin = 80, linear mode
    (1, 16, 16, 400, 160, 1, torch.bfloat16),   # 1x1, Cin = 400 = 12.5 chunks
]


@pytest.mark.parametrize("ng", [0, 1, 2, 3, 4, "wg3", "ks1", "ks2", "pp0", "pp2"])
@pytest.mark.parametrize("case", CASES)
def test_tiled_conv_fwd_epilogue_stats(case, ng, sw):
    """ng: 0 = the library's own choice of tile (half tiles when 256-pixel tiles would under-fill the chip), 1 / 2 = pinned; 4 = the round-5
    experiment: 512-pixel tiles, one workgroup per CU, 16 of 20 accumulator tiles in AccVGPRs, six-stage weight ring (3x3, 160-cout blocks,
    maps up to 95 wide; other geometries keep the library's choice) -- correct, measured slower, opt-in only (profiles/r05_halo_ng4.txt); 3 = the
    other round-5 experiment: 24-row x 16-column patches (384 pixels) for the 96-cout tile on maps whose last patch row hangs over the bottom edge
    by at most 6 % (the 160 x 160 case here) -- correct, no faster, opt-in."""
    ops = _ops()
    from cerberusdet_amd import _lib as L

    sw("CDET_HALO_WG3", None)
    sw("CDET_HALO_KS", None)
    if ng in (3, 4) and not L.load().cdet_has_experiments():
        pytest.skip("the 384- / 512-pixel tiles are compiled into -DCDET_EXPERIMENTS builds of the library only (measured slower, profiles/r05_halo_ng4.txt)")
    if ng in ("pp0", "pp2"):  # 256-pixel tiles on the 4-wave form everywhere / on the 8-wave ping-pong form wherever it fits (csrc/conv_pp.hip); ng = 2
        sw("CDET_CONV_PP", int(ng[2]))  # alone takes the ping-pong form for single-round grids (the default rule)
        sw("CDET_HALO_NG", 2)
    elif ng in ("ks1", "ks2"):  # the half-tile form with its K loop as one chain / split inside the workgroup (conv_halo.hip: KS); other geometries unaffected
        sw("CDET_HALO_KS", int(ng[2]))
        sw("CDET_HALO_NG", int("1"))
    elif ng == "wg3":  # three workgroups per CU for the 96-cout patch form (conv_halo.hip: TRI); other geometries keep the library's choice
        sw("CDET_HALO_WG3", int("1"))
        sw("CDET_HALO_NG", None)
    elif ng:
        sw("CDET_HALO_NG", int(str(ng)))
    else:
        sw("CDET_HALO_NG", None)

    N, H, W, Ci, Co, k, dtype = case
    g = torch.Generator().manual_seed(11)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype)
    w = _rt(torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k), dtype)
    scale = torch.rand(Co, generator=g) + 0.5
    bias = torch.randn(Co, generator=g) * 0.1
    ref_raw = F.conv2d(x, w, None, 1, k // 2)
    res = _rt(torch.randn(N, Co, H, W, generator=g), dtype)
    xb = torch.full((N, H, W, Ci + 16), float("nan"), dtype=dtype, device=DEV)   # neighbours of the slice are NaN: never touched
    xb[..., 8:8 + Ci] = x.permute(0, 2, 3, 1).to(dtype).to(DEV)
    src = ops.View(xb, 8, Ci)
    wf, _ = ops.pack_weight_tiled(w.to(DEV), dtype)
    # raw output + BN partial sums
    dst = ops.new_act(N, H, W, Co, dtype)
    assert ops.conv2d_tiled_ok(src, dst, k, 1)
    nblk = ops.conv_tiled_stat_blocks(src, dst, k)
    stats = torch.zeros(nblk * 2 * Co, device=DEV)
    ops.conv2d_tiled(src, wf, dst, k, stats=stats)
    torch.cuda.synchronize()
    _close(dst.nchw(), ref_raw, 2 ** -7, 1e-3)
    st = stats.view(nblk, 2, Co).sum(0).cpu()
    _close(st[0], ref_raw.sum((0, 2, 3)), 1e-3, 1e-2)
    _close(st[1], (ref_raw ** 2).sum((0, 2, 3)), 1e-3, 1e-2)
    # fused epilogue into a slice of a wider buffer
    yb = torch.full((N, H, W, Co + 16), 7.0, dtype=dtype, device=DEV)
    dsl = ops.View(yb, 8, Co)
    rv = ops.from_nchw(res.to(DEV), dtype)
    ops.conv2d_tiled(src, wf, dsl, k, scale=scale.to(DEV), bias=bias.to(DEV), act=L.ACT_SILU, res=rv)
    torch.cuda.synchronize()
    ref = F.silu(ref_raw * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)) + res
    _close(dsl.nchw(), ref, 2 ** -7, 2e-2)
    assert (yb[..., :8].float() == 7.0).all() and (yb[..., 8 + Co:].float() == 7.0).all(), "conv wrote outside its channel slice"


@pytest.mark.parametrize("case", CASES)
def test_tiled_conv_dgrad_is_forward_on_flipped_operand(case):
    ops = _ops()
    N, H, W, Ci, Co, k, dtype = case
    g = torch.Generator().manual_seed(12)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype).requires_grad_(True)
    w = _rt(torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k), dtype)
    y = F.conv2d(x, w, None, 1, k // 2)
    dy = _rt(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    _, wd = ops.pack_weight_tiled(w.to(DEV), dtype, fwd=False, dgrad=True)
    dyv = ops.from_nchw(dy.to(DEV), dtype)
    dx = ops.new_act(N, H, W, Ci, dtype)
    ops.conv2d_tiled(dyv, wd, dx, k)
    torch.cuda.synchronize()
    _close(dx.nchw(), x.grad, 2 ** -7, 2e-3 * float(x.grad.abs().max()))
    # gradient fan-in: residual = the gradient already there
    prev = ops.from_nchw(_rt(torch.randn(N, Ci, H, W, generator=g), dtype).to(DEV), dtype)
    out = ops.new_act(N, H, W, Ci, dtype)
    ops.conv2d_tiled(dyv, wd, out, k, res=prev)
    torch.cuda.synchronize()
    _close(out.nchw(), x.grad + prev.nchw().float().cpu(), 2 ** -7, 4e-3 * float(x.grad.abs().max()))


PP_CASES = [
    (3, 20, 20, 64, 320, torch.bfloat16),    # linear halo, 5 pixel tiles (odd: the last pair's second group is idle), two cout blocks
    (2, 32, 32, 96, 160, torch.bfloat16),    # 16 x 16 patches, 8 tiles
    (1, 40, 40, 80, 200, torch.float16),     # Cin = 80: partial last chunk; Cout = 200: partial second cout block; 7 tiles
    (5, 16, 16, 32, 160, torch.bfloat16),    # one chunk (9 K steps); patches, 5 tiles (odd)
]


@pytest.mark.parametrize("case", PP_CASES)
def test_ping_pong_form_carries_the_bits_of_the_four_wave_form(case, sw):
    """csrc/conv_pp.hip: the 8-wave two-phase form owns the same 256-pixel tiles / 160-cout blocks and adds in the same K order as
    conv_halo_kernel<.., 9, 5, .., 2>: raw output, BatchNorm partial-sum rows and the fused epilogue (scale, bias, SiLU, residual) must be
    IDENTICAL, incl. an odd number of pixel tiles (the last pair's second group only carries its share of the weight stream), partial chunks
    and partial cout blocks. 20 launches each: the counted waits of the shared weight ring have to hold under repetition."""
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, H, W, Ci, Co, dtype = case
    g = torch.Generator().manual_seed(17)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9), dtype)
    scale, bias = (torch.rand(Co, generator=g) + 0.5).to(DEV), (torch.randn(Co, generator=g) * 0.1).to(DEV)
    src = ops.from_nchw(x.to(DEV), dtype)
    rv = ops.from_nchw(_rt(torch.randn(N, Co, H, W, generator=g), dtype).to(DEV), dtype)
    wf, _ = ops.pack_weight_tiled(w.to(DEV), dtype)
    sw("CDET_HALO_NG", 2)
    out = {}
    for form in (0, 2):
        sw("CDET_CONV_PP", form)
        raw, act = ops.new_act(N, H, W, Co, dtype), ops.new_act(N, H, W, Co, dtype)
        nblk = ops.conv_tiled_stat_blocks(src, raw, 3)
        stats = torch.zeros(nblk * 2 * Co, device=DEV)
        for _ in range(20 if form else 1):
            raw.torch().fill_(7.0)
            stats.fill_(-1.0)
            ops.conv2d_tiled(src, wf, raw, 3, stats=stats)
            ops.conv2d_tiled(src, wf, act, 3, scale=scale, bias=bias, act=L.ACT_SILU, res=rv)
            torch.cuda.synchronize()
            if form:
                assert torch.equal(raw.torch(), out[0][0]) and torch.equal(stats, out[0][1]) and torch.equal(act.torch(), out[0][2])
        out[form] = (raw.torch().clone(), stats.clone(), act.torch().clone())
    ref = F.conv2d(x, w, None, 1, 1)
    _close(out[2][0].permute(0, 3, 1, 2).float().cpu(), ref, 2 ** -7, 1e-3)


def test_tiled_matches_generic_kernel_on_dominant_shape():
    """40x40 320->320 3x3 at batch 4: both kernels see identical inputs; results agree to one output rounding."""
    ops = _ops()
    N, H, W, Ci, Co, k, dtype = 4, 40, 40, 320, 320, 3, torch.bfloat16
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, H, W, Ci, generator=g).to(dtype).to(DEV)
    w = (torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)).to(dtype).float().to(DEV)
    src = ops.View(x)
    a, b = ops.new_act(N, H, W, Co, dtype), ops.new_act(N, H, W, Co, dtype)
    ops.conv2d(src, ops.pack_weight(w, dtype), a, k, 1)
    ops.conv2d_tiled(src, ops.pack_weight_tiled(w, dtype)[0], b, k)
    torch.cuda.synchronize()
    _close(b.torch(), a.torch(), 2 ** -6, 2e-3)


PAIR_CASES = [
    # N, H, W, Cin, Cout, dtype: 1x1 layers with an even number of 160-cout blocks (csrc/conv_pair.hip: two blocks share the pixel tile)
    (3, 17, 13, 96, 640, torch.bfloat16),    # M = 663: tiles straddle images, last tile mostly empty; two pairs
    (1, 12, 12, 416, 320, torch.float16),    # fp16, 13 chunks, one pair
    (2, 20, 20, 80, 320, torch.bfloat16),    # partial last chunk (Cin = 80)
    (1, 40, 40, 1600, 600, torch.bfloat16),  # Cout not a multiple of 160 (4 blocks, the last one 120 wide), 50 chunks
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_pair_kernel_forward_epilogue_stats_and_data_gradient(case, sw):
    """The shared-pixel-tile 1x1 kernel on small shapes (CDET_CONV_PAIR=2 takes it whenever the geometry allows; the library itself uses it
    from 256 workgroups on), through the same entry point cdet_conv2d_tiled: raw output + BN partial sums, fused epilogue into a channel
    slice, and the data gradient (a forward launch on the DGRAD operand) incl. the fan-in form -- against F.conv2d and against the
    4-wave kernel of conv_halo.hip on the same operands."""
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, H, W, Ci, Co, dtype = case
    g = torch.Generator().manual_seed(31)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype).requires_grad_(True)
    w = _rt(torch.randn(Co, Ci, 1, 1, generator=g) / math.sqrt(Ci), dtype)
    ref_raw = F.conv2d(x, w)
    dy = _rt(torch.randn(ref_raw.shape, generator=g), dtype)
    ref_raw.backward(dy)
    ref_raw = ref_raw.detach()
    xb = torch.full((N, H, W, Ci + 16), float("nan"), dtype=dtype, device=DEV)
    xb[..., 8:8 + Ci] = x.detach().permute(0, 2, 3, 1).to(dtype).to(DEV)
    src = ops.View(xb, 8, Ci)
    wf, wd = ops.pack_weight_tiled(w.to(DEV), dtype, fwd=True, dgrad=True)
    scale = torch.rand(Co, generator=g) + 0.5
    bias = torch.randn(Co, generator=g) * 0.1
    res = _rt(torch.randn(N, Co, H, W, generator=g), dtype)
    outs = {}
    for mode in ("2", "0"):
        sw("CDET_CONV_PAIR", int(mode))
        dst = ops.new_act(N, H, W, Co, dtype)
        nblk = ops.conv_tiled_stat_blocks(src, dst, 1)
        stats = torch.zeros(nblk * 2 * Co, device=DEV)
        ops.conv2d_tiled(src, wf, dst, 1, stats=stats)
        yb = torch.full((N, H, W, Co + 16), 7.0, dtype=dtype, device=DEV)
        dsl = ops.View(yb, 8, Co)
        ops.conv2d_tiled(src, wf, dsl, 1, scale=scale.to(DEV), bias=bias.to(DEV), act=L.ACT_SILU, res=ops.from_nchw(res.to(DEV), dtype))
        torch.cuda.synchronize()
        _close(dst.nchw(), ref_raw, 2 ** -7, 1e-3)
        st = stats.view(nblk, 2, Co).sum(0).cpu()
        _close(st[0], ref_raw.sum((0, 2, 3)), 1e-3, 1e-2)
        _close(st[1], (ref_raw ** 2).sum((0, 2, 3)), 1e-3, 1e-2)
        ref = F.silu(ref_raw * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)) + res
        _close(dsl.nchw(), ref, 2 ** -7, 2e-2)
        assert (yb[..., :8].float() == 7.0).all() and (yb[..., 8 + Co:].float() == 7.0).all()
        outs[mode] = (dst.torch().clone(), stats.clone())
    assert torch.equal(outs["2"][0], outs["0"][0])  # same products, same fp32 accumulation order per output: identical bits
    # data gradient: rows = Cin (needs an even number of 160-row blocks to take the pair kernel; otherwise this is the 4-wave kernel)
    sw("CDET_CONV_PAIR", int("2"))
    dyv = ops.from_nchw(dy.to(DEV), dtype)
    dx = ops.new_act(N, H, W, Ci, dtype)
    ops.conv2d_tiled(dyv, wd, dx, 1)
    prev = ops.from_nchw(_rt(torch.randn(N, Ci, H, W, generator=g), dtype).to(DEV), dtype)
    out = ops.new_act(N, H, W, Ci, dtype)
    ops.conv2d_tiled(dyv, wd, out, 1, res=prev)
    torch.cuda.synchronize()
    tol = float(x.grad.abs().max())
    _close(dx.nchw(), x.grad, 2 ** -7, 2e-3 * tol)
    _close(out.nchw(), x.grad + prev.nchw().float().cpu(), 2 ** -7, 4e-3 * tol)


F32_CASES = [
    # N, H, W, Cin, Cout, k, stride, dtype
    (2, 20, 20, 320, 24, 1, 1, torch.bfloat16),    # Detect class projection (nc = 19 / 20 padded to 24) at 20 x 20: half tiles
    (2, 80, 80, 80, 64, 1, 1, torch.bfloat16),     # Detect box projection at 80 x 80
    (1, 40, 40, 320, 24, 1, 1, torch.float16),     # fp16 inference form
    (2, 20, 20, 96, 200, 3, 1, torch.bfloat16),    # 3x3, two cout blocks, linear tiles
    (1, 32, 48, 80, 80, 3, 1, torch.bfloat16),     # 3x3 patch mode, 96-cout tile, half last chunk
    (2, 32, 32, 64, 160, 3, 2, torch.bfloat16),    # stride 2 (conv_vt.hip), patch mode
    (1, 26, 38, 32, 48, 3, 2, torch.float16),      # stride 2, linear tiles
]


@pytest.mark.parametrize("case", F32_CASES)
def test_tiled_conv_fp32_destination_bias_and_accumulate(case):
    """HEPI_F32 epilogue (round 4) of conv_halo.hip / conv_vt.hip: an fp32 destination written straight from the accumulators --
    y = conv * scale + bias into a channel slice of a wider fp32 buffer (Detect's biased 1x1 projections, models/yolo.py:82-100), and the
    accumulate form y += conv (gradient fan-in; the split-operand accuracy chain of tests/hiprec.py). fp32 in, fp32 out: 1e-5 relative
    to the tensor scale (accumulation order only)."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, dtype = case
    g = torch.Generator().manual_seed(41)
    x = _rt(torch.randn(N, Ci, H, W, generator=g), dtype)
    w = _rt(torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k), dtype)
    bias = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, s, k // 2)
    Ho, Wo = ref.shape[2], ref.shape[3]
    src = ops.from_nchw(x.to(DEV), dtype)
    wf, _ = ops.pack_weight_tiled(w.to(DEV), dtype)
    yb = torch.full((N, Ho, Wo, Co + 24), 7.0, dtype=torch.float32, device=DEV)
    dst = ops.View(yb, 8, Co)
    if s == 1:
        assert ops.conv2d_tiled_ok(src, dst, k, 1, accumulate=True)
        run = lambda **kw: ops.conv2d_tiled(src, wf, dst, k, **kw)  # noqa: E731
    else:
        assert ops.conv2d_s2_tiled_ok(src, dst)
        run = lambda **kw: ops.conv2d_s2_tiled(src, wf, dst, **kw)  # noqa: E731
    run(bias=bias.to(DEV))
    torch.cuda.synchronize()
    want = ref + bias.double().view(1, -1, 1, 1)
    scale_ = float(want.abs().max())
    assert float((dst.nchw().double().cpu() - want).abs().max()) <= 1e-5 * scale_
    assert (yb[..., :8] == 7.0).all() and (yb[..., 8 + Co:] == 7.0).all(), "conv wrote outside its channel slice"
    run(accumulate=True)   # y += conv
    run(accumulate=True)
    torch.cuda.synchronize()
    assert float((dst.nchw().double().cpu() - (want + 2 * ref)).abs().max()) <= 1e-5 * 3 * scale_
    assert (yb[..., :8] == 7.0).all() and (yb[..., 8 + Co:] == 7.0).all()
    # a 16-bit destination cannot accumulate: the geometry check refuses it
    d16 = ops.new_act(N, Ho, Wo, Co, dtype)
    if s == 1:
        assert ops.conv2d_tiled_ok(src, d16, k, 1) and not ops.conv2d_tiled_ok(src, d16, k, 1, accumulate=True)


@pytest.mark.parametrize("case", [(2, 16, 16, 160, 64, torch.bfloat16), (1, 13, 19, 48, 32, torch.float16)])
def test_stride2_tiled_dgrad_fp32_destination_accumulates(case):
    """The four-class stride-2 data gradient of conv_vt.hip into an fp32 dX, overwrite then accumulate."""
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, Ho, Wo, Co, Ci, dtype = case
    H, W = 2 * Ho, 2 * Wo
    g = torch.Generator().manual_seed(43)
    w = _rt(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Co * 9), dtype)
    dy = _rt(torch.randn(N, Co, Ho, Wo, generator=g), dtype)
    x = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), None, 2, 1).backward(dy.double())
    ref = x.grad
    dyv = ops.from_nchw(dy.to(DEV), dtype)
    _, wd = ops.pack_weight_tiled(w.to(DEV), dtype, fwd=False, dgrad=True)
    dx = ops.new_act(N, H, W, Ci, torch.float32)
    assert ops.conv2d_s2_tiled_ok(dyv, dx, L.CONV_DGRAD)
    ops.conv2d_s2_tiled_dgrad(dyv, wd, dx)
    ops.conv2d_s2_tiled_dgrad(dyv, wd, dx, accumulate=True)
    torch.cuda.synchronize()
    assert float((dx.nchw().double().cpu() - 2 * ref).abs().max()) <= 1e-5 * 2 * float(ref.abs().max())


CAT_CASES = [
    # N, H, W, parts [(C, upsampled)], Cout, dtype, pair kernel
    (2, 16, 16, [(64, True), (96, False)], 160, torch.bfloat16, False),               # neck top-down: [Upsample(x), backbone tap]
    (1, 20, 24, [(32, False), (64, False), (40, False)], 80, torch.float16, False),    # three parts, last one not a multiple of 32; 96-cout tile
    (3, 8, 8, [(128, True), (64, False)], 320, torch.bfloat16, False),                 # half tiles (few pixels), two cout blocks
    (2, 16, 16, [(640, True), (640, False)], 320, torch.bfloat16, True),               # conv_pair.hip CAT form (YOLOv8x 40 x 40 neck shape)
    (1, 16, 16, [(320, False), (640, False)], 640, torch.float16, True),               # bottom-up: [Conv s2 output, top-down tap], pair form
]


@pytest.mark.parametrize("case", CAT_CASES)
def test_conv_over_virtual_concat_equals_the_materialised_one(case, sw):
    """cdet_conv2d_tiled_cat: the 1x1 convolution behind a Concat (+ Upsample) reading its inputs from their own buffers (slices of wider
    buffers, NaN around them) must give the SAME BITS as cdet_conv2d_tiled on the materialised tensor -- same K order, same kernel body."""
    ops = _ops()
    from cerberusdet_amd import _lib as L

    N, H, W, parts, Co, dtype, pair = case
    sw("CDET_CONV_PAIR", int("2" if pair else "0"))
    g = torch.Generator().manual_seed(71)
    views, mats = [], []
    for Cp, up in parts:
        h, w_ = (H // 2, W // 2) if up else (H, W)
        buf = torch.full((N, h, w_, Cp + 24), float("nan"), dtype=dtype, device=DEV)
        t = _rt(torch.randn(N, h, w_, Cp, generator=g), dtype)
        buf[..., 8:8 + Cp] = t.to(dtype).to(DEV)
        views.append((ops.View(buf, 8, Cp), up))
        mats.append(t.repeat_interleave(2, 1).repeat_interleave(2, 2) if up else t)
    x = torch.cat(mats, 3)
    Ci = x.shape[3]
    w = _rt(torch.randn(Co, Ci, 1, 1, generator=g) / math.sqrt(Ci), dtype)
    scale, bias = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    wf, _ = ops.pack_weight_tiled(w.to(DEV), dtype)
    src = ops.View(x.to(dtype).to(DEV).contiguous())
    want = ops.new_act(N, H, W, Co, dtype)
    ops.conv2d_tiled(src, wf, want, 1, scale=scale.to(DEV), bias=bias.to(DEV), act=L.ACT_SILU)
    yb = torch.full((N, H, W, Co + 16), 7.0, dtype=dtype, device=DEV)
    got = ops.View(yb, 8, Co)
    ops.conv2d_tiled_cat(views, wf, got, scale=scale.to(DEV), bias=bias.to(DEV), act=L.ACT_SILU)
    torch.cuda.synchronize()
    assert torch.equal(got.torch(), want.torch()), f"{int((got.torch() != want.torch()).sum())} of {want.torch().numel()} outputs differ"
    assert (yb[..., :8].float() == 7.0).all() and (yb[..., 8 + Co:].float() == 7.0).all()
    ref = F.silu(F.conv2d(x.permute(0, 3, 1, 2), w) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    _close(got.nchw(), ref, 2 ** -7, 2e-2)

