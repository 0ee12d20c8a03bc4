"""Trainer-level checks on the MI355X: the reference's two-iteration golden (trainers/averaging.py inner loop + optimizer_step
+ EMA, tests/golden/trainer.npz) and the step algebra of the fused path."""
import copy

import numpy as np
import pytest
import torch

import synth
from util import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(meta, mmeta):
    from cerberusdet_amd.models import CerberusDet

    m = CerberusDet(mmeta["tasks"], mmeta["nc"], cfg=copy.deepcopy(mmeta["cfg"]), verbose=False)
    m.sequential_split(mmeta["cfg"]["cerber"], "cpu")
    sd = m.state_dict()
    m.load_state_dict({k: torch.from_numpy(synth.det_tensor(mmeta["seed"], k, v.shape)) for k, v in sd.items()})
    m.hyp = meta["hyp"]
    return m.to(DEV).train()


def test_two_iterations_vs_reference_golden():
    from cerberusdet_amd.trainers import Averaging

    arrays, meta = load_golden("trainer")
    _, mmeta = load_golden("model_tiny2")
    m = _model(meta, mmeta)
    tr = Averaging(torch.device(DEV), m, meta["hyp"], meta["tasks"], epochs=100, nb=1000)
    # parameter groups as the reference's optimizer sees them (bias, decay, bn)
    assert tr.group_sizes == meta["param_group_sizes"]
    assert {str(k): float(max(len(v), 1)) for k, v in tr.serving.items()} == meta["num_branches"]
    sd_before = {k: v.clone() for k, v in m.state_dict().items()}
    for it in range(2):
        batches = {}
        for ti, t in enumerate(meta["tasks"]):
            img = torch.from_numpy(synth.det_image(100 + 10 * it + ti, meta["bs"], meta["imgsz"])).to(DEV)
            b = synth.make_batch(meta["bs"], 2, meta["nc"][ti], 200 + 10 * it + ti)
            batches[t] = dict(img=img, **{k: torch.from_numpy(v).to(DEV) for k, v in b.items()})
        # the reference steps with the constant lr0 here (its scheduler/warm-up is driven by train_epoch, not optimizer_step)
        lrs, mom = [meta["hyp"]["lr0"]] * 3, meta["hyp"]["momentum"]
        out = {t: tr.forward_backward(t, batches[t], active_tasks=meta["tasks"]) for t in meta["tasks"]}
        tr.optimizer_step(lrs, mom)
        torch.cuda.synchronize()
        for t in meta["tasks"]:
            got, want = out[t].cpu().numpy(), arrays[f"it{it}/{t}/items"]
            print(f"it{it} {t}: items {got[:4]} vs reference {want}")
            # bs 2 @64 with random weights is the noise regime of tests/test_oracle_golden.py::test_train_fixture_sensitivity
            assert np.allclose(got[:4], want, rtol=0.15, atol=0.05), (it, t, got, want)
            assert abs(got[4] - 2 * meta["bs"] * got[3]) < 1e-3 * abs(got[4])  # scalar == 2*bs*total (loss.py:179-181)
        sd = m.state_dict()
        esd = tr.ema.ema.state_dict()
        for k in meta["watch"]:
            w_ref, w_got, w0 = arrays[f"it{it}/w/{k}"], sd[k].float().cpu().numpy(), sd_before[k].float().cpu().numpy()
            e_ref, e_got = arrays[f"it{it}/ema/{k}"], esd[k].float().cpu().numpy()
            # the UPDATE (w - w0) is gradient noise-limited; the values themselves must agree to the size of the update
            scale = np.abs(w_ref - synth.det_tensor(mmeta["seed"], k, w_ref.shape)).max() + 1e-12
            assert np.abs(w_got - w_ref).max() < 1.0 * scale + 1e-6, (it, k, np.abs(w_got - w_ref).max(), scale)
            assert np.abs(e_got - e_ref).max() < 1.0 * scale + 1e-6, (it, k)  # d ~ 5e-4 at update 1: the EMA tracks w
            if "running" not in k:
                d_ref, d_got = (w_ref - w0).ravel(), (w_got - w0).ravel()
                if np.linalg.norm(d_ref) > 0:
                    c = float(d_ref @ d_got / (np.linalg.norm(d_ref) * np.linalg.norm(d_got) + 1e-30))
                    print(f"it{it} {k}: update cosine {c:.3f}")
                    assert c > 0.5, (it, k, c)
        # gradients are zeroed by the fused step
        assert all(float(p.grad.abs().max()) == 0.0 for p in m.parameters() if p.grad is not None)


@pytest.mark.parametrize("which", ["model_tiny2", "model_tiny3"])
def test_task_streams_bit_identical_to_sequential_schedule(which):
    """train_step with one HIP stream per task pass (event chain on the shared blocks) must give bit-identical weights, BN running
    statistics, EMA and loss items to the sequential schedule of the reference (trainers/averaging.py:132-194)."""
    from cerberusdet_amd.trainers import Averaging

    arrays, meta = load_golden("trainer")
    _, mmeta = load_golden(which)  # tiny3: three tasks, blocks shared by all of them and by two of them
    tasks, ncs = mmeta["tasks"], mmeta["nc"]
    res = []
    for streams in (False, True):
        m = _model(meta, mmeta)
        tr = Averaging(torch.device(DEV), m, meta["hyp"], tasks, epochs=100, nb=1000, task_streams=streams)
        assert tr.task_streams == streams
        items = []
        for it in range(3):
            batches = {}
            for ti, t in enumerate(tasks):
                img = torch.from_numpy(synth.det_image(500 + 10 * it + ti, 4, 128)).to(DEV)
                b = synth.make_batch(4, 3, ncs[ti], 600 + 10 * it + ti)
                batches[t] = dict(img=img, **{k: torch.from_numpy(v).to(DEV) for k, v in b.items()})
            out = tr.train_step(batches, ni=2000 + it)  # past warm-up: every group has lr > 0
            items.append({t: v.clone() for t, v in out.items()})
        torch.cuda.synchronize()
        res.append((items, {k: v.clone() for k, v in m.state_dict().items()}, {k: v.clone() for k, v in tr.ema.ema.state_dict().items()}))
    (ia, sa, ea), (ib, sb, eb) = res
    for x, y in zip(ia, ib):
        for t in x:
            assert torch.equal(x[t], y[t]), t
    changed = 0
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
        assert torch.equal(ea[k], eb[k]), k
        changed += int(not torch.equal(sa[k].float().cpu(), torch.from_numpy(synth.det_tensor(mmeta["seed"], k, sa[k].shape)).float()))
    assert changed > len(sa) // 2  # the steps really moved the model


def test_freeze_shared_layers_then_unfreeze():
    """--freeze-shared-till-epoch (reference trainers/averaging.py:100-103, cerberus.py:885-925): while frozen, the blocks that serve
    every task keep their weights AND BatchNorm running statistics bit for bit, run from the running statistics (their output equals
    the eval forward), the branches keep learning; after unfreezing they move again and their momentum starts fresh."""
    from cerberusdet_amd.trainers import Averaging

    arrays, meta = load_golden("trainer")
    _, mmeta = load_golden("model_tiny2")
    m = _model(meta, mmeta)
    tr = Averaging(torch.device(DEV), m, meta["hyp"], meta["tasks"], epochs=100, nb=1000, use_ema=False)
    shared = {i for i, ts in tr.serving.items() if len(ts) == len(meta["tasks"])}
    assert shared and len(shared) < len(m.blocks)

    def batches(it):
        out = {}
        for ti, t in enumerate(meta["tasks"]):
            img = torch.from_numpy(synth.det_image(900 + 10 * it + ti, 4, 128)).to(DEV)
            b = synth.make_batch(4, 3, meta["nc"][ti], 950 + 10 * it + ti)
            out[t] = dict(img=img, **{k: torch.from_numpy(v).to(DEV) for k, v in b.items()})
        return out

    def snap():
        return {k: v.clone() for k, v in m.state_dict().items()}

    is_shared = lambda k: int(k.split(".")[1]) in shared  # noqa: E731
    tr.train_step(batches(0), ni=3000)  # one ordinary step first (running statistics move away from their initial values)
    tr.set_shared_frozen(True)
    s0 = snap()
    x = batches(3)[meta["tasks"][0]]["img"]
    plan = m.get_plan(meta["tasks"][0], tuple(x.shape), x.dtype, training=True)
    assert plan.dead == shared and plan.frozen_convs
    for it in (1, 2):
        tr.train_step(batches(it), ni=3000 + it)
    torch.cuda.synchronize()
    s1 = snap()
    for k in s0:
        if "num_batches_tracked" in k:
            continue
        if is_shared(k):
            assert torch.equal(s0[k], s1[k]), k  # weights, biases, BN running statistics: untouched
    assert sum(int(not torch.equal(s0[k], s1[k])) for k in s0 if not is_shared(k) and s0[k].dtype.is_floating_point) > 50
    # the frozen trunk computes the eval forward: its last convolution writes the same activations in the training plan and in an
    # eval plan (same kernels, same folded operands -> bit-identical)
    plan.run_forward(x)
    pe = m.get_plan(meta["tasks"][0], tuple(x.shape), x.dtype, training=False)
    pe.run_forward(x)
    torch.cuda.synchronize()

    def trunk_out(p):
        end = dict(p.fwd_marks)[max(shared)]
        dst = next(args[6] for fn, args in reversed(p.fwd[:end]) if getattr(fn, "__name__", "") == "cdet_conv2d")
        return next(t for t in p.keep if isinstance(t, torch.Tensor) and t.data_ptr() == dst)

    ta, tb = trunk_out(plan), trunk_out(pe)
    assert ta.data_ptr() != tb.data_ptr() and float(ta.float().abs().max()) > 0
    assert torch.equal(ta, tb)
    # unfreeze: the shared blocks learn again
    tr.set_shared_frozen(False)
    tr.train_step(batches(4), ni=3004)
    torch.cuda.synchronize()
    s2 = snap()
    moved = [k for k in s1 if is_shared(k) and s1[k].dtype.is_floating_point and "running" not in k and not torch.equal(s1[k], s2[k])]
    assert len(moved) > 20
    assert all(bool(torch.isfinite(v).all()) for v in s2.values() if v.dtype.is_floating_point)


def test_grad_accumulation_and_block_division():
    """Shared blocks accumulate both tasks' gradients and are divided by 2, branch blocks by 1 (averaging.py:211-217):
    with lr > 0 only for one group we can read the applied update back."""
    from cerberusdet_amd.trainers import Averaging

    arrays, meta = load_golden("trainer")
    _, mmeta = load_golden("model_tiny2")
    m = _model(meta, mmeta)
    tr = Averaging(torch.device(DEV), m, meta["hyp"], meta["tasks"], epochs=100, nb=1000, use_ema=False)
    batches = {}
    for ti, t in enumerate(meta["tasks"]):
        img = torch.from_numpy(synth.det_image(300 + ti, 4, 128)).to(DEV)
        b = synth.make_batch(4, 3, meta["nc"][ti], 400 + ti)
        batches[t] = dict(img=img, **{k: torch.from_numpy(v).to(DEV) for k, v in b.items()})
    for t in meta["tasks"]:
        tr.forward_backward(t, batches[t], active_tasks=meta["tasks"])
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    k_shared, k_branch = "blocks.0.model.1.conv.weight", "blocks.3.cv1.conv.weight"
    g_shared, g_branch = named[k_shared].grad.clone(), named[k_branch].grad.clone()
    w_shared, w_branch = named[k_shared].detach().clone(), named[k_branch].detach().clone()
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
    coef = min(1.0, 10.0 / (total + 1e-6))
    lr = 0.01
    tr.optimizer_step([lr, 0.0, 0.0], 0.0)  # momentum 0 -> update = lr * (clip*g/div + wd*w)
    torch.cuda.synchronize()
    wd = meta["hyp"]["weight_decay"]
    exp_shared = w_shared - lr * (g_shared * coef / 2 + wd * w_shared)
    exp_branch = w_branch - lr * (g_branch * coef / 1 + wd * w_branch)
    assert torch.allclose(named[k_shared], exp_shared, rtol=1e-5, atol=1e-7)
    assert torch.allclose(named[k_branch], exp_branch, rtol=1e-5, atol=1e-7)


def test_sync_bn_path_world1_equals_local_bn():
    """SyncBatchNorm launch list (partials -> sums -> all-reduce -> finalize / apply) through RCCL with a 1-rank group must give
    exactly the per-GPU result (the all-reduce is the identity, counts are multiplied by world = 1)."""
    import os

    import torch.distributed as dist

    from cerberusdet_amd.trainers import Averaging

    arrays, meta = load_golden("trainer")
    _, mmeta = load_golden("model_tiny2")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        res = []
        for sync in (False, True):
            m = _model(meta, mmeta)
            tr = Averaging(torch.device(DEV), m, meta["hyp"], meta["tasks"], epochs=100, nb=1000, use_ema=False, sync_bn=sync)
            t = meta["tasks"][0]
            img = torch.from_numpy(synth.det_image(300, 4, 128)).to(DEV)
            b = synth.make_batch(4, 3, meta["nc"][0], 400)
            out = tr.forward_backward(t, dict(img=img, **{k: torch.from_numpy(v).to(DEV) for k, v in b.items()}), active_tasks=[t])
            torch.cuda.synchronize()
            plan = m.get_plan(t, img.shape, img.dtype, training=True)
            assert plan.sync_bn == sync
            named = dict(m.named_parameters())
            res.append((out.clone(), {k: p.grad.clone() for k, p in named.items() if p.grad is not None},
                        m.state_dict()["blocks.0.model.2.cv1.bn.running_var"].clone()))
        assert torch.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
        assert torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-7)
        for k in res[0][1]:
            # fp32 sums instead of double partial accumulation differ in the last bit; bf16 re-rounding amplifies that down the
            # backward chain (about 1 % at the stem), so compare direction and norm
            a, b_ = res[0][1][k].flatten().double().cpu(), res[1][1][k].flatten().double().cpu()
            if float(a.norm()) < 1e-9:  # branches without positives carry (numerically) no gradient
                continue
            cos = float(a @ b_ / (a.norm() * b_.norm()))
            assert cos > 0.99 and abs(float(b_.norm() / a.norm()) - 1) < 0.05, (k, cos)
    finally:
        dist.destroy_process_group()


def test_overfits_a_fixed_batch():
    """End-to-end sanity of the whole loop (forward, criterion, backward, clip, SGD-Nesterov, EMA): repeating one small batch per task
    must drive both tasks' losses down."""
    from cerberusdet_amd.trainers import Averaging

    arrays, meta = load_golden("trainer")
    _, mmeta = load_golden("model_tiny2")
    m = _model(meta, mmeta)
    hyp = dict(meta["hyp"], lr0=0.01, warmup_epochs=0.0)
    tr = Averaging(torch.device(DEV), m, hyp, meta["tasks"], epochs=100, nb=1000)
    batches = {}
    for ti, t in enumerate(meta["tasks"]):
        img = torch.from_numpy(synth.det_image(40 + ti, 8, 128)).to(DEV)
        b = synth.make_batch(8, 3, meta["nc"][ti], 60 + ti)
        batches[t] = dict(img=img, **{k: torch.from_numpy(v).to(DEV) for k, v in b.items()})
    hist = []
    for it in range(120):
        out = tr.train_step(batches, ni=5000 + it)
        if it % 10 == 0 or it == 119:
            hist.append({t: float(v[3]) for t, v in out.items()})
    print(hist[0], hist[len(hist) // 2], hist[-1])
    for t in meta["tasks"]:
        assert all(np.isfinite(h[t]) for h in hist)
        assert hist[-1][t] < 0.8 * hist[0][t], (t, hist[0][t], hist[-1][t])
