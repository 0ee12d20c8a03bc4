"""Full-size checks on the MI355X at BASELINE.json's configuration (YOLOv8x 2-task, batch 32 per task @640, inference batch 128),
where no CPU oracle finishes in seconds: size-independent properties that pin the kernels exactly.

* convolution: with small-integer operands every product and every partial sum is an exact fp32 integer, so the checksum identity
  sum(y) = <colsum(w), im2col-sums(x)> and the adjoint identities <conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)> hold
  EXACTLY for the layer shapes that dominate the step (stride 1, stride 2 with its parity-class data gradient, 1x1);
* one training pass of the real model: loss identity scalar = 2*bs*total, finite gradients everywhere, a second identical pass
  exactly doubles every accumulated gradient (determinism + accumulate variants);
* NMS at batch 128 x 8400 anchors: descending scores, no surviving same-class pair above the IoU threshold, idempotence;
* assignment at batch 32: every foreground anchor lies inside its ground-truth box, at most top-k anchors per ground truth, scores
  only on foreground anchors and on the assigned class.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (N, H, W, Cin, Cout, k, s): the layers that dominate the iteration + a stride-2 and a 1x1 layer
FULL_CONVS = [(32, 40, 40, 320, 320, 3, 1), (32, 80, 80, 160, 160, 3, 1), (32, 160, 160, 80, 80, 3, 1), (32, 80, 80, 320, 640, 3, 2),
              (32, 40, 40, 1600, 640, 1, 1)]


@pytest.mark.parametrize("case", FULL_CONVS)
def test_conv_checksum_and_adjoint_identities_exact(case):
    from cerberusdet_amd import _lib as L
    from cerberusdet_amd import ops

    N, H, W, Ci, Co, k, s = case
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randint(-2, 3, (N, H, W, Ci), generator=g, device=DEV).to(torch.bfloat16)
    w = torch.randint(-1, 2, (Co, Ci, k, k), generator=g, device=DEV).float()
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    dy = torch.randint(-2, 3, (N, Ho, Wo, Co), generator=g, device=DEV).to(torch.bfloat16)
    xv, dyv = ops.View(x), ops.View(dy)
    # forward into fp32 (exact integers: |y| <= 2 * k*k*Ci)
    y = ops.new_act(N, Ho, Wo, Co, torch.float32)
    ops.conv2d(xv, ops.pack_weight(w, torch.bfloat16), y, k, s)
    # checksum of checksums: sum_p y[p, co] summed over co = sum_k colsum_w[k] * tapsum_x[k]
    xs = x.float().permute(0, 3, 1, 2)
    # tapsum[ci, kh, kw] = sum over the output pixels of the input element that tap reads (zero padding included)
    p = k // 2
    xp = F.pad(xs, (p, p, p, p)).double()
    tsum = torch.empty(Ci, k, k, dtype=torch.float64, device=DEV)
    for kh in range(k):
        for kw in range(k):
            tsum[:, kh, kw] = xp[:, :, kh:kh + s * (Ho - 1) + 1:s, kw:kw + s * (Wo - 1) + 1:s].sum((0, 2, 3))
    want = (w.double().sum(0) * tsum).sum()
    got = y.buf.double().sum()
    assert float(got) == float(want), (float(got), float(want))
    # adjoint identities, all operands exact
    lhs = (y.buf.double() * dy.double()).sum()
    dx = ops.new_act(N, H, W, Ci, torch.float32)
    ops.conv2d(dyv, ops.pack_weight(w, torch.bfloat16, transpose=True), dx, k, s, mode=L.CONV_DGRAD)
    assert float((dx.buf.double() * x.double()).sum()) == float(lhs)
    dw = torch.zeros(Co, Ci, k, k, device=DEV)
    ops.conv2d_wgrad(xv, dyv, dw, k, s)
    assert float((dw.double() * w.double()).sum()) == float(lhs)
    assert float(dw.abs().max()) < 2 ** 24  # the fp32 sums really were exact integers


# (N, H, W, Cin, Cout, k, src slice (ld, coff) or None): every tile form of the tap-resident kernel at BASELINE.json's batch 32 @640 --
# linear 256-pixel tiles (40x40), 16x16 patches (80x80, 160x160 with the 3-fragment / partial-chunk form), half tiles (20x20),
# 1x1 incl. the 160x160 layers that run against HBM, a 2-stage-ring... (none at these widths), and a source / destination that are
# channel slices of C2f / Concat buffers as the engine passes them.
TILED_FULL = [(32, 40, 40, 320, 320, 3, None), (32, 80, 80, 160, 160, 3, None), (32, 160, 160, 80, 80, 3, None),
              (32, 20, 20, 320, 320, 3, None), (32, 80, 80, 320, 320, 3, None), (32, 40, 40, 640, 320, 3, None),
              (32, 80, 80, 960, 320, 1, None), (32, 160, 160, 400, 160, 1, None), (32, 40, 40, 1600, 640, 1, None),
              (32, 160, 160, 160, 160, 1, None), (32, 80, 80, 160, 160, 3, (1280, 320)), (32, 20, 20, 640, 640, 1, (1280, 8))]


def _sparse_pm1(shape, nnz_per_row, row_len, g):
    """Tensor of {-1, 0, +1} whose expected number of non-zeros per reduction row is nnz_per_row (so that |sum| stays far below 256)."""
    p = min(1.0, nnz_per_row / row_len)
    mask = torch.rand(shape, generator=g, device=DEV) < p
    sign = torch.randint(0, 2, shape, generator=g, device=DEV) * 2 - 1
    return (mask * sign).float()


@pytest.mark.parametrize("case", TILED_FULL)
def test_tiled_conv_and_its_data_gradient_exact_at_full_size(case):
    """cdet_conv2d_tiled (csrc/conv_halo.hip: the dominant kernel of the step -- every stride-1 forward and data gradient) at BASELINE
    sizes against the plain fp32 PyTorch reference (tests/torchref.py). Operands are small integers and the weights are sparse enough
    that every output is an integer of magnitude <= 256: exactly representable in bf16, and every fp32 partial sum is exact whatever
    the order -- so the 16-bit output, the BatchNorm partial sums, the data gradient through the flipped / transposed operand and its
    accumulate form (gradient fan-in) must EQUAL the reference bit for bit, over the whole tensor."""
    import torchref as R
    from cerberusdet_amd import ops

    N, H, W, Ci, Co, k, sl = case
    g = torch.Generator(device=DEV).manual_seed(23)
    dtype = torch.bfloat16
    # |x|, |dy| <= 2; weights +-1 with ~48 non-zeros per output row AND per input column -> |y|, |dx| <= 2 * (48 + 6 sigma) < 256
    w = _sparse_pm1((Co, Ci, k, k), 48.0, max(Ci, Co) * k * k, g)
    assert float(w.abs().sum((1, 2, 3)).max()) <= 120 and float(w.abs().sum((0, 2, 3)).max()) <= 120
    ld, coff = sl if sl else (Ci, 0)
    xb = torch.full((N, H, W, ld), float("nan"), dtype=dtype, device=DEV)      # channels outside the slice are NaN: never read
    xb[..., coff:coff + Ci] = torch.randint(-2, 3, (N, H, W, Ci), generator=g, device=DEV).to(dtype)
    src = ops.View(xb, coff, Ci)
    x32 = src.torch().float()
    wf, wd = ops.pack_weight_tiled(w, dtype, fwd=True, dgrad=True)
    # ---- forward: raw output + BN partial sums, into a channel slice when the source is one
    dld, dcoff = (Co + 24, 16) if sl else (Co, 0)
    yb = torch.full((N, H, W, dld), 7.0, dtype=dtype, device=DEV)
    dst = ops.View(yb, dcoff, Co)
    assert ops.conv2d_tiled_ok(src, dst, k, 1)
    nblk = ops.conv_tiled_stat_blocks(src, dst, k)
    stats = torch.zeros(nblk * 2 * Co, device=DEV)
    ops.conv2d_tiled(src, wf, dst, k, stats=stats)
    ref = R.conv_fwd(x32, w, 1)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(dst.torch().float(), ref), f"{int((dst.torch().float() != ref).sum())} of {ref.numel()} outputs differ"
    if sl:
        assert bool((yb[..., :dcoff].float() == 7.0).all()) and bool((yb[..., dcoff + Co:].float() == 7.0).all())
    st = stats.view(nblk, 2, Co).double().sum(0)
    assert torch.equal(st[0], ref.double().sum((0, 1, 2))) and torch.equal(st[1], (ref.double() ** 2).sum((0, 1, 2)))
    del ref
    # ---- data gradient = forward convolution of dY with the DGRAD operand; then the same launch accumulating onto a gradient already there
    dy = torch.randint(-2, 3, (N, H, W, Co), generator=g, device=DEV).to(dtype)
    dyv = ops.View(dy)
    dxb = torch.full((N, H, W, ld), 3.0, dtype=dtype, device=DEV)
    dx = ops.View(dxb, coff, Ci)
    ops.conv2d_tiled(dyv, wd, dx, k)
    dref = R.conv_dgrad(dy.float(), w, 1, H, W)
    assert float(dref.abs().max()) <= 250
    assert torch.equal(dx.torch().float(), dref), f"{int((dx.torch().float() != dref).sum())} of {dref.numel()} gradient elements differ"
    prev = torch.randint(-3, 4, (N, H, W, Ci), generator=g, device=DEV).to(dtype)
    out = ops.new_act(N, H, W, Ci, dtype)
    ops.conv2d_tiled(dyv, wd, out, k, res=ops.View(prev))
    assert torch.equal(out.torch().float(), dref + prev.float())
    # ---- adjoint identity with the (already pinned) weight gradient: <conv(x), dy> = <w, wgrad(x, dy)>, all exact integers
    dw = torch.zeros(Co, Ci, k, k, device=DEV)
    ops.conv2d_wgrad(src, dyv, dw, k, 1)
    assert float((dw.double() * w.double()).sum()) == float((dst.torch().double() * dy.double()).sum())


# (N, H, W of the INPUT, Cin, Cout): the stride-2 rows of YOLOv8x at batch 32 @640 (backbone rows 1, 3, 5, 7 and the two of a neck)
S2_FULL = [(32, 320, 320, 80, 160), (32, 160, 160, 160, 320), (32, 80, 80, 320, 640), (32, 40, 40, 640, 640), (32, 80, 80, 320, 320)]


@pytest.mark.parametrize("case", S2_FULL)
def test_stride2_tiled_conv_and_data_gradient_exact_at_full_size(case):
    """cdet_conv2d_s2_tiled / cdet_conv2d_s2_tiled_dgrad (csrc/conv_vt.hip: parity-plane forward, four-class data gradient) at
    BASELINE sizes against the plain fp32 PyTorch reference, with the small-integer operands of
    test_tiled_conv_and_its_data_gradient_exact_at_full_size: output, BatchNorm partial sums, data gradient and its fan-in form
    must EQUAL the reference bit for bit, and the adjoint identity with the weight gradient holds exactly."""
    import torchref as R
    from cerberusdet_amd import _lib as L
    from cerberusdet_amd import ops

    N, H, W, Ci, Co = case
    Ho, Wo = H // 2, W // 2
    g = torch.Generator(device=DEV).manual_seed(29)
    dtype = torch.bfloat16
    w = _sparse_pm1((Co, Ci, 3, 3), 48.0, max(Ci, Co) * 9, g)
    assert float(w.abs().sum((1, 2, 3)).max()) <= 120 and float(w.abs().sum((0, 2, 3)).max()) <= 120
    x = torch.randint(-2, 3, (N, H, W, Ci), generator=g, device=DEV).to(dtype)
    src = ops.View(x)
    wf, wd = ops.pack_weight_tiled(w, dtype, fwd=True, dgrad=True)
    dst = ops.new_act(N, Ho, Wo, Co, dtype)
    assert ops.conv2d_s2_tiled_ok(src, dst)
    nblk = ops.conv_s2_tiled_stat_blocks(src, dst)
    stats = torch.zeros(nblk * 2 * Co, device=DEV)
    ops.conv2d_s2_tiled(src, wf, dst, stats=stats)
    ref = R.conv_fwd(x.float(), w, 2)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(dst.torch().float(), ref), f"{int((dst.torch().float() != ref).sum())} of {ref.numel()} outputs differ"
    st = stats.view(nblk, 2, Co).double().sum(0)
    assert torch.equal(st[0], ref.double().sum((0, 1, 2))) and torch.equal(st[1], (ref.double() ** 2).sum((0, 1, 2)))
    del ref
    dy = torch.randint(-2, 3, (N, Ho, Wo, Co), generator=g, device=DEV).to(dtype)
    dyv = ops.View(dy)
    dx = ops.new_act(N, H, W, Ci, dtype)
    assert ops.conv2d_s2_tiled_ok(dyv, dx, L.CONV_DGRAD)
    ops.conv2d_s2_tiled_dgrad(dyv, wd, dx)
    dref = R.conv_dgrad(dy.float(), w, 2, H, W)
    assert float(dref.abs().max()) <= 250
    assert torch.equal(dx.torch().float(), dref), f"{int((dx.torch().float() != dref).sum())} of {dref.numel()} gradient elements differ"
    prev = torch.randint(-3, 4, (N, H, W, Ci), generator=g, device=DEV).to(dtype)
    out = ops.View(prev.clone())
    ops.conv2d_s2_tiled_dgrad(dyv, wd, out, res=out)
    assert torch.equal(out.torch().float(), dref + prev.float())
    dw = torch.zeros(Co, Ci, 3, 3, device=DEV)
    ops.conv2d_wgrad(src, dyv, dw, 3, 2)
    assert float((dw.double() * w.double()).sum()) == float((dst.torch().double() * dy.double()).sum())


# BASELINE config 5 (CerberusDetInference: fp16, batch 128 @640): the forward launches with the largest buffers, in the eval form the
# inference plan uses (folded-BN scale / bias in the epilogue, residual for the Bottleneck shortcut), at N = 128.
#   (H, W, Cin, Cout, k, stride, source (ld, coff) or None)
N128_FP16 = [(40, 40, 320, 320, 3, 1, None),          # linear 256-pixel tiles: the most frequent launch of the forward
             (160, 160, 400, 160, 1, 1, (448, 0)),    # C2f cv2 over its 448-pitch concat buffer: 2.9 GB source, 1.3 GB destination
             (160, 160, 80, 80, 3, 1, (448, 160)),    # 16 x 16 patches, 96-cout tile, half last chunk, source = a slice of that buffer
             (40, 40, 1600, 640, 1, 1, None),         # pair tile (conv_pair.hip)
             (320, 320, 80, 160, 3, 2, None)]         # stride 2 on parity planes (conv_vt.hip): 2.1 GB source


@pytest.mark.parametrize("case", N128_FP16)
def test_fp16_batch128_forward_exact_at_full_size(case):
    """The tap-resident forward kernels at BASELINE config 5's size (fp16, N = 128 @640) against the plain fp32 reference
    (tests/torchref.py): small-integer activations, sparse +-1 weights, power-of-two scales and integer biases keep every value an
    exactly representable integer, so raw output, eval epilogue (y * scale + bias, + residual) must match bit for bit over the whole
    tensor -- buffers of 1.3 - 2.9 GB, i.e. byte offsets beyond 2^31 inside the kernels' 32-bit buffer addressing."""
    import torchref as R
    from cerberusdet_amd import ops

    H, W, Ci, Co, k, s, sl = case
    N = 128
    g = torch.Generator(device=DEV).manual_seed(31)
    dtype = torch.float16
    w = _sparse_pm1((Co, Ci, k, k), 40.0, max(Ci, Co) * k * k, g)
    ld, coff = sl if sl else (Ci, 0)
    xb = torch.full((N, H, W, ld), float("nan"), dtype=dtype, device=DEV)
    xb[..., coff:coff + Ci] = torch.randint(-2, 3, (N, H, W, Ci), generator=g, device=DEV).to(dtype)
    src = ops.View(xb, coff, Ci)
    assert xb.numel() * 2 < 3 * 2 ** 30  # (the tiled kernels take sources below 3 GiB; beyond that the plan falls back, see test_c_abi)
    Ho, Wo = H // s, W // s
    wf, _ = ops.pack_weight_tiled(w, dtype, fwd=True, dgrad=False)
    dst = ops.new_act(N, Ho, Wo, Co, dtype)
    ref = R.conv_fwd(src.torch().float(), w, s)
    assert float(ref.abs().max()) <= 200
    if s == 1:
        assert ops.conv2d_tiled_ok(src, dst, k, 1)
        ops.conv2d_tiled(src, wf, dst, k)
    else:
        assert ops.conv2d_s2_tiled_ok(src, dst)
        ops.conv2d_s2_tiled(src, wf, dst)
    assert torch.equal(dst.torch().float(), ref), f"{int((dst.torch().float() != ref).sum())} of {ref.numel()} outputs differ"
    # eval epilogue: scale in {1, 2}, integer bias, residual
    scale = (torch.randint(0, 2, (Co,), generator=g, device=DEV) + 1).float()
    bias = torch.randint(-3, 4, (Co,), generator=g, device=DEV).float()
    res = torch.randint(-2, 3, (N, Ho, Wo, Co), generator=g, device=DEV).to(dtype)
    if s == 1:
        ops.conv2d_tiled(src, wf, dst, k, scale=scale, bias=bias, res=ops.View(res))
    else:
        ops.conv2d_s2_tiled(src, wf, dst, scale=scale, bias=bias, res=ops.View(res))
    want = ref * scale + bias + res.float()
    assert float(want.abs().max()) <= 1024
    assert torch.equal(dst.torch().float(), want)


@pytest.fixture(scope="module")
def v8x_trainer():
    import bench
    from cerberusdet_amd.trainers import Averaging

    dev = torch.device(DEV, 0)
    model, _ = bench.build_model("v8x_2task.yaml", dev)
    tr = Averaging(dev, model, bench.HYP, bench.TASKS, use_ema=False)
    batches = {t: bench.synth_batch(0, ti, 0, 32, [20, 19][ti], 640, dev) for ti, t in enumerate(bench.TASKS)}
    return model, tr, batches


def test_full_model_pass_loss_identity_and_exact_gradient_doubling(v8x_trainer):
    model, tr, batches = v8x_trainer
    import bench

    t = bench.TASKS[0]
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()
    loss5 = tr.forward_backward(t, batches[t], n_max=8, active_tasks=[t])
    torch.cuda.synchronize()
    l = loss5.tolist()
    assert all(math.isfinite(v) for v in l)
    assert abs(l[3] - (l[0] + l[1] + l[2])) <= 1e-5 * abs(l[3])          # total = box + cls + dfl (utils/loss.py:176-181)
    assert abs(l[4] - 2 * 32 * l[3]) <= 1e-5 * abs(l[4])                 # scalar = 2 * bs * total
    named = {k: p for k, p in model.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0}
    assert len(named) >= 180  # every parameter on the task's path received a gradient
    g1 = {k: p.grad.clone() for k, p in named.items()}
    assert all(bool(torch.isfinite(g).all()) for g in g1.values())
    loss5b = tr.forward_backward(t, batches[t], n_max=8, active_tasks=[t])
    torch.cuda.synchronize()
    assert torch.equal(loss5, loss5b)  # train-mode BN uses batch statistics: the second pass is the same computation
    for k, p in named.items():
        assert torch.equal(p.grad, 2 * g1[k]), k  # g + g is exact in fp32: bit-identical kernels, correct accumulate variants


def test_grouped_weight_gradient_launches_equal_per_layer_launches(v8x_trainer, monkeypatch):
    """The engine runs the same-geometry 3x3 weight gradients of a block as ONE grouped launch (Plan._flush_wgrads). Same products, a
    different pixel split -> the fp32 sums differ in their last bits only: every gradient of the real YOLOv8x pass agrees with the
    per-layer launches (CDET_WGRAD_GROUP=0) to 1e-4 of its scale, and the other gradients (data path untouched) exactly."""
    import bench

    model, tr, batches = v8x_trainer
    t = bench.TASKS[1]
    img = batches[t]["img"]

    def one_pass():
        for p in model.parameters():
            if p.grad is not None:
                p.grad.zero_()
        tr.forward_backward(t, batches[t], n_max=8, active_tasks=[t])
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0}

    plan = model.get_plan(t, img.shape, img.dtype, training=True)
    n_grouped = sum(1 for _, cs in plan.bwd_groups for fn, a in cs if getattr(fn, "__name__", "") == "cdet_conv2d_wgrad_grouped")
    assert n_grouped >= 8  # backbone C2f x4 + neck C2f x4 (+ the head's 3x3 convolutions, one geometry each)
    g_grouped = one_pass()
    monkeypatch.setenv("CDET_WGRAD_GROUP", "0")
    model._plans = {}
    g_single = one_pass()
    plan0 = model.get_plan(t, img.shape, img.dtype, training=True)
    assert not any(getattr(fn, "__name__", "") == "cdet_conv2d_wgrad_grouped" for _, cs in plan0.bwd_groups for fn, a in cs)
    monkeypatch.delenv("CDET_WGRAD_GROUP")
    model._plans = {}
    assert g_grouped.keys() == g_single.keys() and len(g_grouped) >= 180
    differ = 0
    for k, a in g_grouped.items():
        b = g_single[k]
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale, k
        differ += int(not torch.equal(a, b))
    assert differ <= 80  # only the grouped 3x3 convolution weights may differ at all (72 stride-1 3x3 layers on a task's path)


def test_three_task_full_size_iteration_streams_vs_sequential():
    """BASELINE config 4's model (YOLOv8x, three tasks: blocks shared by all three and by two of them) at full width on one GPU: two
    iterations with one HIP stream per task (block-interleaved enqueue, grouped weight gradients) give bit-identical loss items, weights
    and BatchNorm statistics to the sequential schedule; the loss identity scalar = 2*bs*total holds for every task; every block's
    weights moved. Batch 8 per task keeps it short -- the kernels and launch lists are the full-size ones."""
    import yaml

    import bench
    from cerberusdet_amd.models import CerberusDet
    from cerberusdet_amd.trainers import Averaging

    dev = torch.device(DEV, 0)
    tasks, ncs = ["voc", "objects365_animals", "objects365_tableware"], [20, 19, 12]
    cfg = yaml.safe_load(open(bench.ROOT / "cerberusdet_amd" / "models" / "cfg" / "v8x_3task.yaml"))
    hyp = dict(bench.HYP)
    for k in ("box", "cls", "dfl"):
        if isinstance(hyp[k], (list, tuple)):
            hyp[k] = list(hyp[k]) + [hyp[k][-1]] * (3 - len(hyp[k]))
    data = {t: bench.synth_batch(0, i, 0, 8, ncs[i], 640, dev) for i, t in enumerate(tasks)}
    res = []
    for streams in (False, True):
        torch.manual_seed(0)
        m = CerberusDet(tasks, ncs, cfg=cfg, verbose=False)
        m.sequential_split(cfg["cerber"], "cpu")
        m.hyp = hyp
        m = m.to(dev).train()
        w0 = {k: v.clone() for k, v in m.state_dict().items()}
        tr = Averaging(dev, m, hyp, tasks, use_ema=False, task_streams=streams)
        assert tr.task_streams == streams
        items = [tr.train_step(data, n_max=8, ni=3000 + it) for it in range(2)]
        torch.cuda.synchronize()
        tr.check_targets()
        res.append(([{t: v.clone() for t, v in it.items()} for it in items], {k: v.clone() for k, v in m.state_dict().items()}))
        for it in items:
            for t in tasks:
                l = it[t].tolist()
                assert all(math.isfinite(v) for v in l) and abs(l[4] - 2 * 8 * l[3]) <= 1e-5 * abs(l[4])
        moved = {k.split(".")[1] for k, v in m.state_dict().items() if k.endswith("conv.weight") and not torch.equal(v, w0[k])}
        assert moved == {str(i) for i in range(len(m.blocks)) if any(True for _ in m.blocks[i].parameters())}
        del tr, m
    (ia, sa), (ib, sb) = res
    for x, y in zip(ia, ib):
        for t in tasks:
            assert torch.equal(x[t], y[t]), t
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_nms_full_batch_properties():
    import bench
    from cerberusdet_amd import ops

    pred = bench.nms_inputs(128, 20, 8400).to(DEV)
    rows, cnt = ops.nms_batched(pred, 0.25, 0.45, max_det=300)
    torch.cuda.synchronize()
    rows_h, cnt_h = rows.cpu(), cnt.cpu().tolist()
    assert all(0 < c <= 300 for c in cnt_h)
    for i in (0, 17, 127):
        r = rows_h[i, :cnt_h[i]]
        assert bool((r[1:, 4] <= r[:-1, 4]).all())  # descending confidence
        assert bool((r[:, 4] > 0.25).all())
        b = r[:, :4] + r[:, 5:6] * 7680.0  # class offsets as in general.py:462
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt, rb = torch.max(b[:, None, :2], b[None, :, :2]), torch.min(b[:, None, 2:], b[None, :, 2:])
        inter = (rb - lt).clamp(0).prod(2)
        iou = inter / (area[:, None] + area[None] - inter)
        iou.fill_diagonal_(0)
        assert float(iou.max()) <= 0.45  # no surviving pair of one class above the threshold
    # idempotence: the survivors, fed back as the only candidates, all survive in the same order
    i = 5
    k = cnt_h[i]
    r = rows[i, :k]
    again = torch.zeros((1, 24, 8400), dtype=torch.float32, device=DEV)
    again[0, 0, :k], again[0, 1, :k] = (r[:, 0] + r[:, 2]) / 2, (r[:, 1] + r[:, 3]) / 2
    again[0, 2, :k], again[0, 3, :k] = r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]
    again[0, 4 + r[:, 5].long(), torch.arange(k, device=DEV)] = r[:, 4]
    rows2, cnt2 = ops.nms_batched(again, 0.25, 0.45, max_det=300)
    assert int(cnt2[0]) == k
    assert torch.equal(rows2[0, :k, 4:], r[:, 4:]) and torch.allclose(rows2[0, :k, :4], r[:, :4], atol=1e-3)


def test_assignment_full_batch_properties():
    import bench
    from cerberusdet_amd import ops
    from cerberusdet_amd.utils.loss import pad_targets

    N, nc, topk = 32, 20, 10
    g = torch.Generator(device=DEV).manual_seed(3)
    feats = [torch.randn((N, 640 // s, 640 // s, 64 + 24), generator=g, device=DEV) * 0.5 for s in (8, 16, 32)]
    for f in feats:
        f[..., 64:64 + nc] -= 4.0  # realistic low class logits
    batch = bench.synth_batch(0, 0, 1, N, nc, 640, torch.device(DEV, 0))
    gt = pad_targets(batch, N, (640, 640), torch.device(DEV, 0), n_max=8)
    loss5, dfe, asg = ops.det_loss(feats, gt, nc, dict(box=7.5, cls=0.5, dfl=1.5), (8, 16, 32), want_assign=True, topk=topk)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss5).all()) and all(bool(torch.isfinite(d).all()) for d in dfe)
    fg, gi = asg["fg_mask"].bool(), asg["target_gt_idx"].long()
    # anchor centres in pixels, level by level (utils/tal.py:181-193)
    pts = []
    for s in (8, 16, 32):
        n = 640 // s
        yy, xx = torch.meshgrid(torch.arange(n, device=DEV) + 0.5, torch.arange(n, device=DEV) + 0.5, indexing="ij")
        pts.append(torch.stack((xx.flatten(), yy.flatten()), 1) * s)
    pts = torch.cat(pts, 0)
    assert int(fg.sum()) > 0
    for b in range(N):
        idx = torch.nonzero(fg[b]).flatten()
        box = gt[b, gi[b, idx], 1:5]
        c = pts[idx]
        assert bool(((c[:, 0] > box[:, 0]) & (c[:, 0] < box[:, 2]) & (c[:, 1] > box[:, 1]) & (c[:, 1] < box[:, 3])).all())
        per_gt = torch.bincount(gi[b, idx], minlength=gt.shape[1])
        assert int(per_gt.max()) <= topk
        assert bool((asg["target_labels"][b, idx].long() == gt[b, gi[b, idx], 0].long()).all())
    ts = asg["target_scores"]
    assert float(ts[~fg].abs().max()) == 0.0
    on_label = torch.zeros_like(ts, dtype=torch.bool).scatter_(2, asg["target_labels"].long().clamp(0, nc - 1).unsqueeze(-1), True)
    assert float(ts[~on_label].abs().max()) == 0.0
