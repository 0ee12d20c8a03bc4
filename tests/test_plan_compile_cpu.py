"""Host-side checks that need no GPU: the plan compiler (engine.Plan) on the shipped model YAMLs, and the fp32 GEMM-form convolution
reference used by the GPU parity tests.

engine.Plan only CALLS kernels when a plan runs; compiling one needs the library's host-side geometry helpers (cdet_conv2d_tiled_ok,
*_stat_blocks, *_ws_elems) and torch allocations, both of which work on the CPU. So every (task set, train / eval) combination the
reference's forward accepts (cerberus.py:804-882: a str, a list, None = all tasks; model.train() or model.eval()) is compiled here."""
import os

import pytest
import torch
import torch.nn.functional as F
import yaml

import torchref as R

CFG_DIR = os.path.join(os.path.dirname(__file__), "..", "cerberusdet_amd", "models", "cfg")
NCS = {"voc": 20, "objects365_animals": 19, "objects365_tableware": 12}


def _model(cfg_name, tasks):
    from cerberusdet_amd.models import CerberusDet

    cfg = yaml.safe_load(open(os.path.join(CFG_DIR, cfg_name)))
    m = CerberusDet(tasks, [NCS[t] for t in tasks], cfg=cfg, verbose=False)
    if cfg.get("cerber"):
        m.sequential_split(cfg["cerber"], "cpu")
    return m


@pytest.mark.parametrize("cfg_name,tasks", [("v8x_2task.yaml", ["voc", "objects365_animals"]),
                                            ("v8x_3task.yaml", ["voc", "objects365_animals", "objects365_tableware"]),
                                            ("v8n_2task.yaml", ["voc", "objects365_animals"])])
def test_every_task_set_compiles_in_train_and_eval_form(cfg_name, tasks):
    from cerberusdet_amd.engine import Plan

    m = _model(cfg_name, tasks)
    dev = torch.device("cpu")
    single = {}
    for training in (False, True):
        for ts in [[t] for t in tasks] + [tasks]:
            p = Plan(m, ts, 2, 64, 64, training, torch.bfloat16, torch.uint8, dev)
            assert p.n_fwd_calls > 0 and (p.n_bwd_calls > 0) == training
            if len(ts) == 1:
                single[(training, ts[0])] = p
            elif training:
                # the multi-task training plan (model.train(); model(x)): the shared trunk runs once, every task's branch is there,
                # and each conv unit has a backward -- fewer launches than the per-task plans together
                convs = sum(1 for r in p.trace if r["kind"] == "conv")
                per_task = [sum(1 for r in single[(True, t)].trace if r["kind"] == "conv") for t in tasks]
                assert max(per_task) < convs < sum(per_task)
                assert all("bwd_lo" in r for r in p.trace if r["kind"] in ("conv", "bias"))
                assert any(r["kind"] == "copy" for r in p.trace)  # the backbone taps both necks concatenate are copied, not placed
    # single-task training plans keep every Concat input in place except the ones a non-Conv producer feeds
    assert sum(1 for r in single[(True, tasks[0])].trace if r["kind"] == "copy") <= 2


def test_eval_plan_fuses_the_first_two_backbone_rows(monkeypatch):
    """Eval plans run backbone rows 0 and 1 as one launch (csrc/stem_conv1.hip: the stem's map stays in LDS); training plans and
    CDET_STEM_FUSE=0 keep the two-kernel form."""
    from cerberusdet_amd.engine import Plan

    tasks = ["voc", "objects365_animals"]
    m = _model("v8x_2task.yaml", tasks)
    dev = torch.device("cpu")

    def names(plan):
        return [getattr(fn, "__name__", "") for fn, _ in plan.fwd]

    fused = Plan(m, tasks, 2, 64, 64, False, torch.bfloat16, torch.uint8, dev)
    assert names(fused).count("cdet_stem_conv1") == 1 and "cdet_stem_conv" not in names(fused)
    assert fused.fwd[0][1] is fused.img_slots[0] and fused.fwd[0][1][9:14] == [2, 64, 64, 80, 160]
    monkeypatch.setenv("CDET_STEM_FUSE", "0")
    plain = Plan(m, tasks, 2, 64, 64, False, torch.bfloat16, torch.uint8, dev)
    assert "cdet_stem_conv1" not in names(plain) and names(plain).count("cdet_stem_conv") == 1
    assert plain.n_fwd_calls == fused.n_fwd_calls + 1
    monkeypatch.delenv("CDET_STEM_FUSE")
    train = Plan(m, [tasks[0]], 2, 64, 64, True, torch.bfloat16, torch.uint8, dev)
    assert "cdet_stem_conv1" not in names(train)


def test_eval_plan_keeps_concat_and_upsample_virtual(monkeypatch):
    """Eval plans never build the neck's Concat / Upsample tensors: the C2f cv1 behind each Concat reads its inputs itself
    (cdet_conv2d_tiled_cat). CDET_VCAT=0 is the round-3 form with upsample / copy launches."""
    from cerberusdet_amd.engine import Plan

    tasks = ["voc", "objects365_animals"]
    m = _model("v8x_2task.yaml", tasks)
    dev = torch.device("cpu")

    def names(plan):
        return [getattr(fn, "__name__", "") for fn, _ in plan.fwd]

    v = names(Plan(m, tasks, 2, 64, 64, False, torch.bfloat16, torch.uint8, dev))
    assert v.count("cdet_conv2d_tiled_cat") == 8 and "cdet_upsample2" not in v and "cdet_copy_channels" not in v  # 4 neck Concats per task
    monkeypatch.setenv("CDET_VCAT", "0")
    r3 = names(Plan(m, tasks, 2, 64, 64, False, torch.bfloat16, torch.uint8, dev))
    assert "cdet_conv2d_tiled_cat" not in r3 and r3.count("cdet_upsample2") >= 2 and r3.count("cdet_copy_channels") >= 2
    assert len(r3) == len(v) + r3.count("cdet_upsample2") + r3.count("cdet_copy_channels")
    monkeypatch.delenv("CDET_VCAT")
    tr = names(Plan(m, [tasks[0]], 2, 64, 64, True, torch.bfloat16, torch.uint8, dev))
    assert "cdet_conv2d_tiled_cat" not in tr


def test_frozen_trunk_plan_compiles_without_backward_for_shared_blocks():
    from cerberusdet_amd.engine import Plan
    from cerberusdet_amd.models import CerberusDet

    tasks = ["voc", "objects365_animals"]
    m = _model("v8n_2task.yaml", tasks)
    CerberusDet.freeze_shared_layers(m)
    frozen = tuple(i for i, b in enumerate(m.blocks) if any(True for _ in b.parameters()) and not any(p.requires_grad for p in b.parameters()))
    assert 0 in frozen
    p = Plan(m, [tasks[0]], 2, 64, 64, True, torch.bfloat16, torch.uint8, torch.device("cpu"), frozen=frozen)
    assert 0 in p.dead and all(idx not in p.dead or not calls for idx, calls in p.bwd_groups)


@pytest.mark.parametrize("case", [(2, 9, 7, 5, 6, 3, 1), (2, 8, 8, 4, 3, 3, 2), (1, 5, 6, 7, 2, 1, 1), (2, 7, 9, 3, 4, 3, 2)])
def test_gemm_form_reference_equals_torch_conv2d_and_its_autograd(case):
    N, H, W, Ci, Co, k, s = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g)
    xn = x.permute(0, 3, 1, 2).clone().requires_grad_()
    wn = w.clone().requires_grad_()
    y = F.conv2d(xn, wn, None, s, k // 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dyn = dy.permute(0, 2, 3, 1).contiguous()
    assert torch.allclose(R.conv_fwd(x, w, s), y.detach().permute(0, 2, 3, 1), atol=1e-4)
    assert torch.allclose(R.conv_dgrad(dyn, w, s, H, W), xn.grad.permute(0, 2, 3, 1), atol=1e-4)
    assert torch.allclose(R.conv_wgrad(x, dyn, k, s), wn.grad, atol=1e-4)
    # integer operands: exact, whatever the order
    xi, wi = torch.randint(-2, 3, x.shape).float(), torch.randint(-1, 2, w.shape).float()
    assert torch.equal(R.conv_fwd(xi, wi, s), F.conv2d(xi.permute(0, 3, 1, 2), wi, None, s, k // 2).permute(0, 2, 3, 1))


def test_model_with_compiled_plans_deep_copies_without_runtime_state():
    """ModelEMA / branch cloning deep-copy the model while launch lists (ctypes descriptors), packed operands
    and the trainer's IPC exchange hang off its modules: none of that may travel (or break the copy)."""
    import copy

    from cerberusdet_amd.engine import Plan

    tasks = ["voc", "objects365_animals"]
    m = _model("v8n_2task.yaml", tasks)
    dev = torch.device("cpu")
    Plan(m, tasks, 2, 64, 64, False, torch.bfloat16, torch.uint8, dev)
    Plan(m, [tasks[0]], 2, 64, 64, True, torch.bfloat16, torch.uint8, dev)
    m._peer_xchg = object()
    c = copy.deepcopy(m)
    assert not hasattr(c, "_peer_xchg") and m._peer_xchg is not None
    assert not any(a in mod.__dict__ for mod in c.modules() for a in type(m)._RUNTIME_ATTRS)
    assert any("_plan_slots" in mod.__dict__ for mod in m.modules())  # the original keeps its runtime state
    assert c.state_dict().keys() == m.state_dict().keys()


def test_training_plan_marks_where_each_backbone_row_is_complete_and_release_unregisters():
    """Round 5: block 0's backward carries one mark per backbone row (engine.Plan.bwd_sub) -- where the trainer folds and all-reduces that row's
    slice of the trunk's gradient bucket while the backward runs on (reference: DDP's bucketed overlap, train.py:182-184); the grouped weight
    gradients of a row are flushed AT its mark, not at the end of the block. An evicted plan takes its argument lists off the modules."""
    from cerberusdet_amd.engine import Plan

    tasks = ["voc", "objects365_animals"]
    m = _model("v8x_2task.yaml", tasks)
    dev = torch.device("cpu")
    p = Plan(m, [tasks[0]], 2, 64, 64, True, torch.bfloat16, torch.uint8, dev)
    rows = p.bwd_sub[0]
    n_rows = len(m.blocks[0].model)
    assert [r for r, _ in rows] == list(range(n_rows - 1, -1, -1))        # rows finish from the last to the first
    ends = [e for _, e in rows]
    calls0 = dict(p.bwd_groups)[0]
    assert ends == sorted(ends) and len(set(ends)) == n_rows and ends[-1] == len(calls0)
    names = [getattr(fn, "__name__", "") for fn, _ in calls0]
    lo = 0
    for row, end in rows:                                                  # a C2f row's grouped weight gradient is its segment's last launch
        seg = names[lo:end]
        if "cdet_conv2d_wgrad_grouped" in seg:
            assert seg[-1] == "cdet_conv2d_wgrad_grouped" and type(m.blocks[0].model[row]).__name__ == "C2f"
        lo = end
    assert names.count("cdet_conv2d_wgrad_grouped") >= 3
    assert set(p.bwd_sub) == {0}
    ev = Plan(m, tasks, 2, 64, 64, False, torch.bfloat16, torch.uint8, dev)
    assert not ev.bwd_sub
    before = sum(len(mod.__dict__.get("_plan_slots", [])) for mod in m.modules())
    ev.release()
    after = sum(len(mod.__dict__.get("_plan_slots", [])) for mod in m.modules())
    assert after < before and after == sum(1 for _ in p._registered)
