"""Teacher-forced, launch-by-launch parity of the compiled TRAIN plans on the MI355X (tests/teacher.py): every convolution unit,
head projection, shortcut, upsample and concat copy of the launch list is compared with an fp32 PyTorch evaluation of that single
layer from the engine's own buffers -- the tap-resident kernels (csrc/conv_halo.hip, conv_wgrad_halo.hip), the in-place Concat
slices, the grouped weight gradients and the accumulate variants included, on

  * a half-width YOLOv8 2-task model (channels 32 ... 256, cerber split), single-task plans as the trainer runs them AND the 2-task
    training plan `model(x)` compiles in train mode (reference cerberus.py:804-882);
  * the real YOLOv8x at BASELINE.json's configuration (batch 32 @640): every unit of one task's plan.
"""
import copy

import pytest
import torch

import synth
import teacher
from util import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _half_width_model():
    from cerberusdet_amd.models import CerberusDet

    _, meta0 = load_golden("model_tiny2")
    cfg = copy.deepcopy(meta0["cfg"])
    cfg["width_multiple"], cfg["depth_multiple"] = 0.5, 0.33
    m = CerberusDet(meta0["tasks"], meta0["nc"], cfg=copy.deepcopy(cfg), verbose=False)
    m.sequential_split(cfg["cerber"], "cpu")
    m.load_state_dict({k: torch.from_numpy(synth.det_tensor(9, k, v.shape)) for k, v in m.state_dict().items()})
    return m.to(DEV).train(), meta0


def _gt(bs, n, nc, imgsz, seed):
    g = torch.Generator().manual_seed(seed)
    cls = torch.randint(0, nc, (bs, n, 1), generator=g).float()
    cxy = (torch.rand(bs, n, 2, generator=g) * 0.6 + 0.2) * imgsz
    wh = (torch.rand(bs, n, 2, generator=g) * 0.3 + 0.05) * imgsz
    return torch.cat((cls, cxy - wh / 2, cxy + wh / 2), 2).contiguous().to(DEV)


def _zero_grads(model):
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()


def _run(plan, img, tasks, ncs, seed=5):
    rep = teacher.Report()
    _zero_grads(plan.model)
    plan.run_forward(img)
    torch.cuda.synchronize()
    teacher.check_forward(plan, rep)
    for ti, t in enumerate(tasks):
        plan.loss(t, _gt(img.shape[0], 4, ncs[ti], img.shape[2], seed + ti), dict(box=7.5, cls=0.5, dfl=1.5))
    n = teacher.check_backward(plan, rep)
    torch.cuda.synchronize()
    return rep, n


@pytest.mark.parametrize("which", ["task0", "task1", "both"])
def test_half_width_train_plan_every_launch_vs_fp32_layer(which):
    m, meta = _half_width_model()
    tasks = {"task0": meta["tasks"][:1], "task1": meta["tasks"][1:2], "both": list(meta["tasks"])}[which]
    ncs = [meta["nc"][meta["tasks"].index(t)] for t in tasks]
    img = torch.from_numpy(synth.det_image(31, 8, 128)).to(DEV)
    plan = m.get_plan(tasks, img.shape, img.dtype, training=True)
    kinds = [r["kind"] for r in plan.trace]
    assert kinds.count("conv") >= 50 and kinds.count("bias") == 6 * len(tasks) and "up" in kinds
    assert any(r.get("also") is not None for r in plan.trace), "Bottleneck shortcuts: their gradient sum rides on cv2's BatchNorm backward"
    if which == "both":  # a backbone tap that both tasks' necks concatenate is copied, not placed (overwrite / accumulate bookkeeping)
        assert "copy" in kinds
    names = {getattr(fn, "__name__", "") for _, cs in plan.bwd_groups for fn, _ in cs} | {getattr(fn, "__name__", "") for fn, _ in plan.fwd}
    assert {"cdet_conv2d_tiled", "cdet_conv2d_tiled_dgrad", "cdet_conv2d_wgrad"} <= names
    # round 6: the head projections' data gradients moved to the tap-resident kernel too -- the round-1 generic kernel no longer appears in a train plan
    # (it stays the fallback for geometries the tiled kernels refuse and is pinned on its own in tests/test_gpu_kernels.py)
    assert "cdet_conv2d" not in names and all(r.get("dgrad_tiled") for r in plan.trace if r["kind"] == "bias")
    rep, n = _run(plan, img, tasks, ncs)
    print(f"[teacher/half-width/{which}] {n} backward units, {len(rep.rows)} tensors: {rep.summary()}")
    assert n >= 60


@pytest.mark.parametrize("ti", [0, 1])
def test_v8x_full_size_train_plan_every_launch_vs_fp32_layer(ti):
    """YOLOv8x, batch 32 @640 (BASELINE.json config 2), each task's training plan (nc = 20 and nc = 19: the second head pads its class
    channels to 24): 97 Conv units + 6 head projections, each pinned."""
    import bench

    dev = torch.device(DEV, 0)
    model, _ = bench.build_model("v8x_2task.yaml", dev)
    t = bench.TASKS[ti]
    batch = bench.synth_batch(0, ti, 0, 32, bench.NC[ti], 640, dev)
    img = batch["img"]
    plan = model.get_plan(t, img.shape, img.dtype, training=True)
    names = [getattr(fn, "__name__", "") for _, cs in plan.bwd_groups for fn, _ in cs]
    assert names.count("cdet_conv2d_wgrad_grouped") >= 8 and names.count("cdet_conv2d_tiled_dgrad") >= 60
    assert sum(1 for fn, _ in plan.fwd if getattr(fn, "__name__", "") in ("cdet_conv2d_tiled", "cdet_conv2d_tiled_bn")) >= 80
    rep, n = _run(plan, img, [t], [bench.NC[ti]])
    print(f"[teacher/v8x bs32@640 {t}] {n} backward units, {len(rep.rows)} tensors: {rep.summary()}")
    assert sum(1 for r in plan.trace if r["kind"] == "conv") == 97 and n >= 104
    assert sum(1 for r in plan.trace if r.get("also") is not None) == 18  # the backbone's Bottleneck shortcuts (3 + 6 + 6 + 3)
