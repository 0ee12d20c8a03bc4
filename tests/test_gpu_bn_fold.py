"""BatchNorm partial sums reduced inside the launch that produces them (csrc/bn_fold.h, round 5) against the two-kernel form
(reference semantics: nn.BatchNorm2d in train mode behind every convolution, models/common.py:57-62; eps 1e-3, momentum 0.03,
utils/torch_utils.py:184-186).

The in-launch form hands data between workgroups on different CUs / XCDs inside one kernel (write-through partial rows, ticket counters, the
last arrivers add in a fixed tree order), so it is tested the way MI355X_MICROARCH.md asks for such hand-offs: thousands of launches, every word
compared, inputs that change from launch to launch (a stale row of the previous launch would show), geometries from one cluster of rows up to
the 3200 workgroups of the 160 x 160 layers at batch 32 that spread over all XCDs, with and without another stream loading the chip. Equality is
`torch.equal` on the partial rows, the fp32 totals, mean, invstd and the running statistics (forward); the sums, dgamma, dbeta (backward).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from cerberusdet_amd import _lib as L
from cerberusdet_amd import ops
from cerberusdet_amd.ops import View, conv_desc, dt

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPS, MOM = 1e-3, 0.03


@pytest.fixture(autouse=True)
def _needs_experiments_build():
    """Round 6: the fold lost its A/B (profiles/r05_bn_fold_ab.txt) and left the product build -- its code is compiled into
    `make -C cerberusdet_amd/csrc EXTRA=-DCDET_EXPERIMENTS` builds only; the product library refuses a fold descriptor (asserted below)."""
    if not L.load().cdet_has_experiments():
        pytest.skip("the in-launch BatchNorm fold is compiled into -DCDET_EXPERIMENTS builds of the library only")


class Fold:
    """Plan-side resources of the fold (engine.Plan._fold_desc / _bind_folds in miniature)."""

    def __init__(self):
        self.tickets = torch.zeros(L.BN_FOLD_TICKET_WORDS, dtype=torch.int32, device=DEV)
        self.keep = []

    def desc(self, nrows, Cn, **ptrs):
        lib = L.load()
        d = L.BnFold()
        d.nrows, d.C, d.ncl = nrows, Cn, -(-nrows // 32)
        cl = torch.empty(int(lib.cdet_bn_fold_cl_doubles(nrows, Cn)), dtype=torch.float64, device=DEV)
        d.tickets, d.cl_sums = self.tickets.data_ptr(), cl.data_ptr()
        for k, v in ptrs.items():
            setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        dev = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(DEV)
        self.keep += [cl, dev, d]
        return dev


# (name, N, H, W, Cin, Cout, k, stride): one cluster, partial clusters, several cout blocks, the pair tile (1x1, K >= 640, 640 couts), the
# narrow 96-cout tile, stride 2 (conv_vt), and the 160 x 160 layer at batch 32: 3200 rows = 100 clusters over all 8 XCDs
GEOMS = [
    ("one-cluster 20x20 64->64", 2, 20, 20, 64, 64, 3, 1),
    ("partial 20x20 320->320", 3, 20, 20, 320, 320, 3, 1),
    ("40x40 320->320 bs8", 8, 40, 40, 320, 320, 3, 1),
    ("40x40 1x1 640->640 (pair tile)", 32, 40, 40, 640, 640, 1, 1),
    ("80x80 1x1 320->160", 4, 80, 80, 320, 160, 1, 1),
    ("80x80 160->160 bs8", 8, 80, 80, 160, 160, 3, 1),
    ("160x160 80->80 bs32 (3200 rows)", 32, 160, 160, 80, 80, 3, 1),
    ("s2 160x160->80x80 80->160", 8, 160, 160, 80, 160, 3, 2),
    ("s2 40x40->20x20 320->640", 16, 40, 40, 320, 640, 3, 2),
]


def _conv_case(N, H, W, Ci, Co, k, s, seed):
    g = torch.Generator().manual_seed(seed)
    xs = [View((torch.randn(N, H, W, Ci, generator=g) * 0.7 + 0.1 * j).to(torch.bfloat16).to(DEV)) for j in range(3)]
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).to(DEV)
    wt = ops.pack_weight_tiled(w, torch.bfloat16)[0]
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    z = View(torch.empty(N, Ho, Wo, Co, dtype=torch.bfloat16, device=DEV))
    return xs, wt, z


@pytest.mark.parametrize("geom", GEOMS, ids=[g[0] for g in GEOMS])
def test_forward_fold_equals_conv_plus_bn_finalize_over_many_launches(geom):
    name, N, H, W, Ci, Co, k, s = geom
    lib = L.load()
    xs, wt, z = _conv_case(N, H, W, Ci, Co, k, s, seed=len(name))
    d = conv_desc(xs[0], z, k, s)
    if s == 1:
        assert lib.cdet_conv2d_tiled_ok(C.byref(d)) and lib.cdet_conv2d_tiled_bn_ok(C.byref(d)), name
        nblk = lib.cdet_conv2d_tiled_stat_blocks(C.byref(d))
        plain, folded = lib.cdet_conv2d_tiled, lib.cdet_conv2d_tiled_bn
    else:
        assert lib.cdet_conv2d_s2_tiled_ok(C.byref(d)) and lib.cdet_conv2d_s2_tiled_bn_ok(C.byref(d)), name
        nblk = lib.cdet_conv2d_s2_tiled_stat_blocks(C.byref(d))
        plain, folded = lib.cdet_conv2d_s2_tiled, lib.cdet_conv2d_s2_tiled_bn
    M = z.M
    st = torch.cuda.current_stream().cuda_stream
    fold = Fold()
    f32 = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=DEV)  # noqa: E731
    stats_a, stats_b = f32(nblk * 2 * Co), f32(nblk * 2 * Co)
    mean_a, inv_a, mean_b, inv_b, totals = f32(Co), f32(Co), f32(Co), f32(Co), f32(2 * Co)
    rm_a, rv_a, rm_b, rv_b = f32(Co), torch.ones(Co, device=DEV), f32(Co), torch.ones(Co, device=DEV)
    fd = fold.desc(nblk, Co, totals=totals, mean=mean_b, invstd=inv_b, running_mean=rm_b, running_var=rv_b, inv_count=1.0 / M,
                   unbias=M / (M - 1), eps=EPS, momentum=MOM)
    # another stream keeps the chip busy during every second batch of launches (uneven load: tickets and rows arrive in other orders)
    side = torch.cuda.Stream()
    big = torch.randn(1 << 27, device=DEV)
    n_launch = 240 if nblk <= 400 else 120
    for it in range(n_launch):
        x = xs[it % 3]
        if it % 40 == 20:
            with torch.cuda.stream(side):
                for _ in range(6):
                    big.mul_(1.0001)
        L.check(plain(C.byref(d), x.buf.data_ptr(), wt.data_ptr(), None, None, None, z.buf.data_ptr(), stats_a.data_ptr(), st), "conv")
        L.check(lib.cdet_bn_finalize(stats_a.data_ptr(), nblk, Co, M, EPS, MOM, rm_a.data_ptr(), rv_a.data_ptr(), mean_a.data_ptr(), inv_a.data_ptr(), st), "fin")
        za = z.buf.clone()
        L.check(folded(C.byref(d), x.buf.data_ptr(), wt.data_ptr(), z.buf.data_ptr(), stats_b.data_ptr(), fd.data_ptr(), st), "conv_bn")
        if it % 8 == 7 or it < 4:
            torch.cuda.synchronize()
            assert torch.equal(z.buf, za), (name, it, "raw output")
            assert torch.equal(stats_a, stats_b), (name, it, "partial rows")
            assert torch.equal(mean_a, mean_b) and torch.equal(inv_a, inv_b), (name, it, "mean / invstd")
            assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b), (name, it, "running statistics")
            assert int(fold.tickets.abs().sum()) == 0, (name, it, "tickets are reset by the last arrivers")
            # the totals are what the SyncBatchNorm list would all-reduce: finalize(totals, nblk = 1) gives the same mean / invstd
            m2, i2 = f32(Co), f32(Co)
            L.check(lib.cdet_bn_finalize(totals.data_ptr(), 1, Co, M, EPS, MOM, None, None, m2.data_ptr(), i2.data_ptr(), st), "fin1")
            torch.cuda.synchronize()
            assert torch.equal(m2, mean_b) and torch.equal(i2, inv_b), (name, it, "totals")
    torch.cuda.synchronize()
    # and the numbers mean something: against torch on the kernel's own output
    zf = z.buf.float().reshape(-1, Co)
    assert float(((mean_b - zf.mean(0)).abs() / (zf.std(0) + 1e-6)).max()) < 2e-3
    assert float((inv_b * torch.sqrt(zf.var(0, unbiased=False) + EPS) - 1).abs().max()) < 2e-3


BWD = [(2, 20, 20, 64), (8, 40, 40, 320), (32, 80, 80, 160), (32, 160, 160, 80), (3, 24, 24, 24)]


@pytest.mark.parametrize("shape", BWD, ids=[f"{n}x{h}x{w}x{c}" for n, h, w, c in BWD])
def test_backward_fold_equals_reduce_plus_bn_bwd_sums_over_many_launches(shape):
    N, H, W, Cn = shape
    lib = L.load()
    g = torch.Generator().manual_seed(Cn)
    M = N * H * W
    zs = [View((torch.randn(N, H, W, Cn, generator=g) * (1 + j)).to(torch.bfloat16).to(DEV)) for j in range(3)]
    dys = [View(torch.randn(N, H, W, Cn, generator=g).to(torch.bfloat16).to(DEV)) for j in range(3)]
    mean, invstd = (torch.randn(Cn, generator=g) * 0.1).to(DEV), (torch.rand(Cn, generator=g) + 0.5).to(DEV)
    gamma, beta = (torch.rand(Cn, generator=g) + 0.5).to(DEV), (torch.randn(Cn, generator=g) * 0.1).to(DEV)
    nb = lib.cdet_bn_bwd_blocks(M)
    st = torch.cuda.current_stream().cuda_stream
    part_a = torch.zeros(nb * 2 * Cn + 2 * Cn, device=DEV)
    part_b = torch.zeros_like(part_a)
    dg_a, db_a, dg_b, db_b = (torch.zeros(Cn, device=DEV) for _ in range(4))
    fold = Fold()
    sums_b = part_b[nb * 2 * Cn:]
    fd = fold.desc(nb, Cn, totals=sums_b.data_ptr(), dgamma=dg_b, dbeta=db_b, accumulate=1)
    args = lambda dy, z, part: (dy.buf.data_ptr(), dy.ld, 0, z.buf.data_ptr(), z.ld, 0, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),  # noqa: E731
                                beta.data_ptr(), part.data_ptr(), M, Cn, dt(torch.bfloat16))
    for it in range(300):
        dy, z = dys[it % 3], zs[(it // 3) % 3]
        L.check(lib.cdet_bn_silu_bwd_reduce(*args(dy, z, part_a), st), "reduce")
        L.check(lib.cdet_bn_bwd_sums(part_a.data_ptr(), nb, Cn, part_a[nb * 2 * Cn:].data_ptr(), dg_a.data_ptr(), db_a.data_ptr(), 1, st), "sums")
        L.check(lib.cdet_bn_silu_bwd_reduce_fold(*args(dy, z, part_b), fd.data_ptr(), st), "reduce_fold")
        if it % 10 == 9 or it < 3:
            torch.cuda.synchronize()
            assert torch.equal(part_a, part_b), (shape, it, "partial rows + sums")
            assert torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b), (shape, it, "dgamma / dbeta (accumulated over the launches)")
            assert int(fold.tickets.abs().sum()) == 0
    # the apply pass takes the folded sums in its nblk = 0 form: same dz as the two-launch form with its own reduction
    dz_a = View(torch.empty(N, H, W, Cn, dtype=torch.bfloat16, device=DEV))
    dz_b = View(torch.empty(N, H, W, Cn, dtype=torch.bfloat16, device=DEV))
    dy, z = dys[0], zs[0]
    L.check(lib.cdet_bn_silu_bwd_reduce(*args(dy, z, part_a), st), "reduce")
    L.check(lib.cdet_bn_silu_bwd_apply(dy.buf.data_ptr(), dy.ld, 0, z.buf.data_ptr(), z.ld, 0, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                       beta.data_ptr(), part_a.data_ptr(), nb, dg_a.data_ptr(), db_a.data_ptr(), 1, dz_a.buf.data_ptr(), dz_a.ld, 0, M, Cn,
                                       dt(torch.bfloat16), 0, st), "apply")
    L.check(lib.cdet_bn_silu_bwd_reduce_fold(*args(dy, z, part_b), fd.data_ptr(), st), "reduce_fold")
    L.check(lib.cdet_bn_silu_bwd_apply(dy.buf.data_ptr(), dy.ld, 0, z.buf.data_ptr(), z.ld, 0, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                       beta.data_ptr(), sums_b.data_ptr(), 0, None, None, 1, dz_b.buf.data_ptr(), dz_b.ld, 0, M, Cn,
                                       dt(torch.bfloat16), 0, st), "apply(nblk = 0)")
    torch.cuda.synchronize()
    assert torch.equal(dz_a.buf, dz_b.buf) and torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b)


def test_running_update_batched_equals_per_layer_finalize():
    """cdet_bn_running_update (the later task's deferred running-statistics updates of a shared block, one launch per block) against
    cdet_bn_finalize(totals, nblk = 1) per layer."""
    lib = L.load()
    g = torch.Generator().manual_seed(4)
    st = torch.cuda.current_stream().cuda_stream
    layers = []
    for Cn, count in ((80, 819200), (160, 204800), (320, 51200), (640, 12800), (24, 7)):
        tot = torch.stack((torch.randn(Cn, generator=g) * count * 0.1, torch.rand(Cn, generator=g) * count + count)).reshape(-1).to(DEV)
        rm, rv = torch.randn(Cn, generator=g).to(DEV), (torch.rand(Cn, generator=g) + 0.5).to(DEV)
        layers.append((Cn, count, tot, rm, rv, rm.clone(), rv.clone()))
    tab = (L.BnRunningItem * len(layers))()
    for it, (Cn, count, tot, rm, rv, rm2, rv2) in zip(tab, layers):
        it.totals, it.running_mean, it.running_var = tot.data_ptr(), rm2.data_ptr(), rv2.data_ptr()
        it.inv_count, it.unbias, it.momentum, it.C = 1.0 / count, count / (count - 1), MOM, Cn
    tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(DEV)
    for _ in range(3):
        for Cn, count, tot, rm, rv, rm2, rv2 in layers:
            m_, i_ = torch.empty(Cn, device=DEV), torch.empty(Cn, device=DEV)
            L.check(lib.cdet_bn_finalize(tot.data_ptr(), 1, Cn, count, EPS, MOM, rm.data_ptr(), rv.data_ptr(), m_.data_ptr(), i_.data_ptr(), st), "fin")
        L.check(lib.cdet_bn_running_update(tab_dev.data_ptr(), len(layers), 640, st), "upd")
    torch.cuda.synchronize()
    for Cn, count, tot, rm, rv, rm2, rv2 in layers:
        assert torch.equal(rm, rm2) and torch.equal(rv, rv2), Cn


def _train_two_iterations(monkeypatch, fold):
    import copy

    import synth
    from cerberusdet_amd.models import CerberusDet
    from cerberusdet_amd.trainers import Averaging
    from util import load_golden

    monkeypatch.setenv("CDET_BN_FOLD", "1" if fold else "0")
    _, meta0 = load_golden("model_tiny2")
    cfg = copy.deepcopy(meta0["cfg"])
    cfg["width_multiple"], cfg["depth_multiple"] = 0.5, 0.33
    m = CerberusDet(meta0["tasks"], meta0["nc"], cfg=copy.deepcopy(cfg), verbose=False)
    m.sequential_split(cfg["cerber"], "cpu")
    m.load_state_dict({k: torch.from_numpy(synth.det_tensor(9, k, v.shape)) for k, v in m.state_dict().items()})
    m = m.to(DEV).train()
    hyp = load_golden("trainer")[1]["hyp"]
    m.hyp = hyp
    tr = Averaging(torch.device(DEV), m, hyp, meta0["tasks"], epochs=100, nb=1000)
    items = None
    names = set()
    for it in range(2):
        batches = {}
        for ti, t in enumerate(meta0["tasks"]):
            b = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_batch(8, 3, meta0["nc"][ti], 50 + 7 * it + ti).items()}
            b["img"] = torch.from_numpy(synth.det_image(70 + 3 * it + ti, 8, 128)).to(DEV)
            batches[t] = b
        items = tr.train_step(batches, n_max=8, ni=2000 + it)
    torch.cuda.synchronize()
    n_fin = n_fold = 0
    for plan in m._plans.values():
        if plan.training:
            calls = [getattr(fn, "__name__", "") for fn, _ in plan.fwd] + [getattr(fn, "__name__", "") for _, cs in plan.bwd_groups for fn, _ in cs]
            calls += [getattr(fn, "__name__", "") for cs in plan.deferred_stats.values() for fn, _ in cs]
            names |= set(calls)
            n_fin += calls.count("cdet_bn_finalize")
            n_fold += calls.count("cdet_conv2d_tiled_bn") + calls.count("cdet_conv2d_s2_tiled_bn")
    return ({t: v.cpu().numpy() for t, v in items.items()}, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()},
            {k: v.detach().cpu().numpy() for k, v in tr.ema.ema.state_dict().items()}, names, n_fin, n_fold)


def test_training_iterations_bit_identical_with_and_without_the_fold(monkeypatch):
    """Two iterations of the 2-task trainer (task streams, decoupled passes: the later task's running statistics travel as ONE deferred launch per
    shared block) with the fold on and off: loss items, every weight, every running statistic and the EMA are bit-identical; with the fold the
    plans hold no bn_finalize launch except for the layers the fold does not take (the 3-channel stem)."""
    it1, sd1, ema1, names1, n_fin1, n_fold1 = _train_two_iterations(monkeypatch, True)
    it0, sd0, ema0, names0, n_fin0, n_fold0 = _train_two_iterations(monkeypatch, False)
    assert n_fold0 == 0 and n_fold1 >= 100 and "cdet_bn_silu_bwd_reduce_fold" in names1 and "cdet_bn_silu_bwd_reduce_fold" not in names0
    assert "cdet_bn_running_update" in names1 and n_fin1 <= 4 < n_fin0, (n_fin1, n_fin0)
    for t in it0:
        assert np.array_equal(it0[t], it1[t]), t
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    for k in ema0:
        assert np.array_equal(ema0[k], ema1[k]), k
