"""The workload the headline numbers are quoted on, CHECKED at its own size (round 5): the YOLOv8x 2-task all-heads EVAL plan at
batch 32 @640 bf16 (BASELINE.json north_star: the forward the 40 % MFMA-roofline target is timed on) and at batch 128 @640 fp16
(BASELINE.json configs[4]: CerberusDetInference) -- the round-4 eval forms (virtual Concat / Upsample sources of cdet_conv2d_tiled_cat incl. the
(y/2, x/2) upsampled segment, the fused first two backbone rows csrc/stem_conv1.hip, the fp32-destination projections, the one-launch SPPF pool
chain, early head lanes, fresh output sets) had only been compared with anything at N <= 3, H <= 128.

  * teacher-forced, launch by launch (tests/teacher.py::check_eval_forward): every unit of the compiled plan against an fp32 evaluation of that
    single layer from the engine's OWN input buffers -- 2^-7 of the tensor scale for 16-bit outputs, 1e-3 for fp32 maps and boxes;
  * bit-identity of `y` and the head maps at full size against the round-3 forms (CDET_VCAT=0, CDET_STEM_FUSE=0, CDET_EARLY_HEADS=0,
    CDET_EVAL_LANES=0, and all of them at once), and of the default call's fresh output tensors against the plan's own;
  * CerberusDetInference.predict at batch 128: the rows it keeps for 4 images against oracle/nms.py applied to the downloaded `y`.

Reference: cerberusdet/models/yolo.py:87-100 (Detect.forward eval branch), models/common.py:51-68, 174-191, 230-245, 288-295,
cerberusdet/cerberusdet_inference.py:85-186.
"""
import numpy as np
import pytest
import torch

import teacher

pytestmark = pytest.mark.gpu
DEV = "cuda"

CONFIGS = {"bs32_bf16": (32, torch.bfloat16), "bs128_fp16": (128, torch.float16)}


def _model(dtype):
    import bench

    dev = torch.device(DEV, 0)
    model, _ = bench.build_model("v8x_2task.yaml", dev)
    model.eval()
    (model.half if dtype == torch.float16 else model.bfloat16)()
    # random-init BatchNorm running statistics are (0, 1): give every layer a distinct folded scale / bias so that a swapped or stale
    # epilogue vector cannot go unnoticed (deterministic; keeps activations O(1))
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_((torch.rand(m.running_mean.shape, generator=g) * 0.2 - 0.1).to(DEV))
                m.running_var.copy_((torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75).to(DEV))
                m.weight.copy_((torch.rand(m.weight.shape, generator=g) * 0.4 + 0.8).to(DEV))
                m.bias.copy_((torch.rand(m.bias.shape, generator=g) * 0.2 - 0.1).to(DEV))
    model.mark_weights_changed()
    return model


def _image(bs, dtype):
    return torch.rand(bs, 3, 640, 640, generator=torch.Generator().manual_seed(3)).to(dtype).to(DEV)


def _drop_plans(model):
    for p in model._plans.values():
        p.release()
    model._plans = {}
    torch.cuda.empty_cache()


def _outputs(model, x, **kw):
    with torch.no_grad():
        out = model(x, **kw)
    torch.cuda.synchronize()
    return {t: (y.clone(), [f.clone() for f in maps]) for t, (y, maps) in out.items()}


def _same(a, b):
    return all(torch.equal(a[t][0], b[t][0]) and all(torch.equal(p, q) for p, q in zip(a[t][1], b[t][1])) for t in a)


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_v8x_full_size_eval_plan_every_launch_vs_fp32_layer(cfg):
    """Every Conv unit of the 2-task plan, the 12 projections, the SPPF pool chain and both decodes, at the size the
    north-star forward (bf16, batch 32) and config 5 (fp16, batch 128) run at, exactly as compiled by default: virtual Concat, fused stem,
    tiled fp32 projections, stream lanes."""
    import bench

    bs, dtype = CONFIGS[cfg]
    model = _model(dtype)
    x = _image(bs, dtype)
    tasks = list(bench.TASKS)
    plan = model.get_plan(tasks, x.shape, x.dtype, training=False)
    names = [getattr(fn, "__name__", "") for fn, _ in plan.fwd]
    # the round-4 forms are what is being checked
    assert names.count("cdet_stem_conv1") == 1 and names.count("cdet_conv2d_tiled_cat") == 8 and "cdet_upsample2" not in names
    assert names.count("cdet_sppf_pool") == 1 and names.count("cdet_detect_decode") == 2 and plan.sched is not None
    assert len(plan.fwd) == 157
    plan.fresh_outputs(False)
    with torch.no_grad():
        plan.run_forward(x)
    torch.cuda.synchronize()
    rep = teacher.Report()
    n = teacher.check_eval_forward(plan, rep)
    print(f"[teacher/eval v8x {cfg}] {n}: {rep.summary()}")
    # every Conv module on the plan's path is checked exactly once (the fused first two backbone rows as one unit): the shared trunk once, each
    # task's neck and head once
    assert n["econv"] + 2 * n["stemc1"] == len(plan.convs) and len(plan.convs) == 143  # (143 BatchNorm layers in the 2-task model)
    assert n["stemc1"] == 1 and n["ebias"] == 12 and n["epool"] == 1 and n["decode"] == 2
    assert sum(1 for r in plan.trace if r["kind"] == "econv" and hasattr(r["x"], "parts")) == 8
    assert sum(1 for r in plan.trace if r["kind"] == "econv" and hasattr(r["x"], "parts") and any(up for _, up in r["x"].parts)) == 4
    assert sum(1 for r in plan.trace if r["kind"] == "econv" and r["res"] is not None) == 18  # the backbone's Bottleneck shortcuts
    for t in tasks:  # nothing degenerate went through the checks
        y = plan.y[t].float()
        assert bool(torch.isfinite(y).all()) and float(y[:, :4].std()) > 1.0 and float(y[:, 4:].std()) > 0


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_v8x_full_size_eval_outputs_bit_identical_across_plan_forms(cfg, monkeypatch):
    """`y` and the six head maps of the default plan at full size against the round-3 forms of every round-4 change, one at a time and all
    together; and the default call's fresh tensors against the plan-owned ones."""
    bs, dtype = CONFIGS[cfg]
    model = _model(dtype)
    x = _image(bs, dtype)
    ref = _outputs(model, x)                       # default call: fresh output tensors (engine.Plan.fresh_outputs)
    again = _outputs(model, x)
    home = _outputs(model, x, zero_copy=True)
    assert _same(ref, again), "two default forwards of the same input differ"
    assert _same(ref, home), "the fresh output set differs from the plan-owned one"
    with torch.no_grad():
        a, b = model(x), model(x)
    assert all(a[t][0].data_ptr() != b[t][0].data_ptr() for t in a), "default calls must hand out fresh tensors (reference cerberus.py:804-882)"
    del a, b
    switches = ["CDET_VCAT", "CDET_STEM_FUSE", "CDET_EARLY_HEADS", "CDET_EVAL_LANES"]
    for off in [[s] for s in switches] + [switches]:
        for s in off:
            monkeypatch.setenv(s, "0")
        _drop_plans(model)
        got = _outputs(model, x)
        plan = next(iter(model._plans.values()))
        names = [getattr(fn, "__name__", "") for fn, _ in plan.fwd]
        if "CDET_VCAT" in off:
            assert "cdet_conv2d_tiled_cat" not in names and "cdet_upsample2" in names
        if "CDET_STEM_FUSE" in off:
            assert "cdet_stem_conv1" not in names and "cdet_stem_conv" in names
        if "CDET_EVAL_LANES" in off:
            assert plan.sched is None
        assert _same(ref, got), f"outputs with {off} = 0 differ from the default plan's"
        for s in off:
            monkeypatch.delenv(s)
    _drop_plans(model)


def test_predict_batch_128_rows_equal_oracle_nms_on_the_downloaded_outputs():
    """CerberusDetInference.predict (reference cerberusdet_inference.py:85-186) at BASELINE config 5's size -- fp16, batch 128 @640, both tasks,
    class biases shifted until ~100 detections per image survive: the result dicts of 4 images (first, two in the middle, last) must equal the CPU
    oracle's NMS + cross-task merge + scale_boxes (oracle/nms.py) applied to the `y` tensors downloaded from the same forward."""
    import copy

    import bench
    from cerberusdet_amd.cerberusdet_inference import CerberusDetInference
    from oracle import nms as on

    model = _model(torch.float16)
    det = CerberusDetInference(copy.deepcopy(model), device="cuda:0", half=True, img_size=640)
    del model
    x = _image(128, torch.float16)
    n_res = bench.calibrate_detections(det, x, (720, 1280))
    assert n_res >= 20, f"calibration left {n_res} detections per image"
    res = det.predict(x, original_shape=(720, 1280))
    with torch.no_grad():
        out = det.model(x, zero_copy=True)
    torch.cuda.synchronize()
    pick = [0, 41, 86, 127]
    # fp16 `y` as it is: the reference filters and converts xywh -> xyxy in the input dtype and promotes to fp32 behind that (general.py:446-449)
    y_cpu = {t: y[pick].cpu().numpy() for t, (y, _) in out.items()}
    assert all(v.dtype == np.float16 for v in y_cpu.values())
    want = on.predict_postprocess(y_cpu, det.names, (640, 640), (720, 1280), conf_thres=det.conf_thres, iou_thres=det.iou_thres,
                                  iou_thres_between_tasks=det.iou_thres_between_tasks, max_det=300)
    assert len(res) == 128
    total = 0
    for i, w in zip(pick, want):
        got = res[i]
        assert len(got) == len(w), (i, len(got), len(w))
        for a, b in zip(got, w):
            assert a["box"] == b["box"] and a["label"] == b["label"] and a["task"] == b["task"] and a["label_name"] == b["label_name"], (i, a, b)
            assert np.float32(a["score"]) == np.float32(b["score"]), (i, a, b)
        total += len(w)
    assert total >= 80 and len({d["task"] for i in pick for d in res[i]}) == 2
    print(f"[predict bs128] {n_res:.1f} detections per image, {total} rows of 4 images equal the oracle's")


@pytest.mark.parametrize("bs", [32, 128])
def test_v8x_fp16_eval_plan_meets_1e3_of_the_full_precision_forward_at_full_width(bs):
    """BASELINE.json north_star asks for >= 40 % of the MFMA peak AND boxes / loss within 1e-3 rel of the reference. The bf16 plans meet the first and
    miss the second end to end (1.1 - 1.5 px of 640 = 2e-3: seven mantissa bits); the fp16 plans -- the reference's own inference dtype
    (cerberusdet_inference.py:34-40 `model.half()`, models/yolo.py:87-100) -- run the same kernels at the same MFMA rate with three more bits. ASSERTED
    here at full width (YOLOv8x 2-task, randomised BatchNorm statistics so that every layer is O(1)), at the north-star batch 32 and at config 5's 128,
    against model.full_precision() (the fp32 reference's numbers to ~1e-6, tests/test_gpu_full_precision.py) on four images of the batch:
    boxes <= 1e-3 x 640 px, class probabilities <= 1e-3 absolute. bench.py times this very plan as `north_star_fwd_fp16`."""
    model = _model(torch.float16)
    x = _image(bs, torch.float16)
    with torch.no_grad():
        out = model(x)  # the full batch: the plan of the timed workload
        lo = {t: y[:4].float().clone() for t, (y, maps) in out.items()}
        del out
        model.full_precision()
        hi = model(x[:4].contiguous())
    torch.cuda.synchronize()
    box = max(float((lo[t][:, :4] - hi[t][0][:, :4]).abs().max()) for t in hi)
    prob = max(float((lo[t][:, 4:] - hi[t][0][:, 4:]).abs().max()) for t in hi)
    var = min(float((b - b.mean(dim=(0, 2, 3), keepdim=True)).abs().max()) for t in hi for b in hi[t][1])  # the part of a head map the network computes
    wh = min(float(hi[t][0][:, 2:4].max() - hi[t][0][:, 2:4].min()) for t in hi)
    print(f"[v8x fp16 bs{bs} vs full precision] boxes max |d| {box:.3f} px = {box / 640:.2e} of the image, class probabilities max |d| {prob:.2e} "
          f"(smallest head-map variation {var:.3f}, box sizes spread over {wh:.1f} px)")
    assert box <= 1e-3 * 640, box
    assert prob <= 1e-3, prob
    assert var > 1e-2 and wh > 1.0  # nothing degenerate was compared
    model.half()


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_v8x_16_bit_eval_plan_against_the_full_precision_forward_of_the_same_weights(cfg):
    """Whole-model accuracy of the 16-bit eval plans at full width, on non-degenerate activations (the randomised BatchNorm statistics of _model keep
    every layer O(1)): the north-star plan (bf16) and config 5's (fp16) against model.full_precision() (cerberusdet_amd/precise.py -- the fp32
    reference's numbers to ~1e-6, tests/test_gpu_full_precision.py) on four images of the batch. Measured: bf16 head-map error 7.2 % of the maps' variation (rel-L2), boxes
    1.1 px; fp16 (three more mantissa bits) 0.89 %, 0.23 px -- a factor 8 apart, as storage noise must be. Asserted at 1.5x those."""
    bs, dtype = CONFIGS[cfg]
    model = _model(dtype)
    x = _image(bs, dtype)[:4].contiguous()
    with torch.no_grad():
        lo = {t: (y.float().clone(), [f.float().clone() for f in maps]) for t, (y, maps) in model(x).items()}
    model.full_precision()
    hi = model(x)
    torch.cuda.synchronize()
    # (the head maps are dominated by their biases -- bias_init puts the class logits near -10 --: the error is measured against the part of a map that
    #  the network computes, i.e. relative to the map's variation around its per-channel mean)
    def centred(b):
        return b - b.mean(dim=(0, 2, 3), keepdim=True)

    rel = max(float((a - b).norm() / centred(b).norm()) for t in hi for a, b in zip(lo[t][1], hi[t][1]))
    box = max(float((lo[t][0][:, :4] - hi[t][0][:, :4]).abs().max()) for t in hi)
    var = min(float(centred(b).abs().max()) for t in hi for b in hi[t][1])
    print(f"[v8x {cfg} vs full precision] head maps: error / variation (rel-L2) {rel:.4f}, boxes max |d| {box:.3f} px (smallest map variation {var:.3f})")
    bands = (0.11, 3.2) if dtype == torch.bfloat16 else (0.014, 0.5)   # measured 0.072 / 1.10 px and 0.0089 / 0.23 px: 16-bit storage noise, 8x apart
    assert rel <= bands[0] and box <= bands[1], (rel, box)
    assert var > 1e-2   # nothing degenerate was compared
    (model.half if dtype == torch.float16 else model.bfloat16)()
