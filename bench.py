#!/usr/bin/env python3
"""Headline benchmark: images/sec of the YOLOv8x 2-task TRAINING step @640 (BASELINE.json configs[1]) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one iteration of the reference's trainer (trainers/averaging.py:132-223): for each of the 2 tasks a batch of 32
synthetic 640x640 images -> forward -> TAL/CIoU/DFL/BCE loss -> backward (gradients accumulate) ; then gradient all-reduce
(N > 1), global-norm clip, per-block division, SGD-Nesterov, EMA. bf16 storage, fp32 accumulation and master weights.
Inputs are resident in HBM before the timed region (synthetic data). Rank 0 prints ONE JSON line.
"""
import argparse
import math
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the two task passes run on their own HIP streams; with the default of 4 hardware queues RCCL's streams push both onto ONE queue
# (measured: 114.5 ms per step with 4, 101.6 ms with 8) -- must be set before the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TASKS = ["voc", "objects365_animals"]
NC = [20, 19]
HYP = dict(box=[7.5, 7.5], cls=[0.5, 0.5], dfl=[1.5, 1.5], lr0=0.00309, lrf=0.0956, momentum=0.952, weight_decay=0.00037,
           warmup_epochs=2.04, warmup_momentum=0.898, warmup_bias_lr=0.0502)
GFLOP_FWD_PER_IMG_TASK = 257.48  # conv FLOPs of one task's path @640 (README.md:237)
MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, MI355X_MICROARCH.md


def synth_batch(rank, t, i, bs, nc, imgsz, device, boxes_per_img=8):
    """uint8 images from seed 1000+97r+13t+i; 8 boxes/img, cls~U, cxcy~U(.2,.8), wh~U(.05,.35)."""
    g = torch.Generator().manual_seed(1000 + 97 * rank + 13 * t + i)
    img = torch.randint(0, 256, (bs, 3, imgsz, imgsz), dtype=torch.uint8, generator=g)
    n = bs * boxes_per_img
    cls = torch.randint(0, nc, (n, 1), generator=g).float()
    cxy = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(n, 2, generator=g) * 0.3 + 0.05
    bi = torch.arange(bs).repeat_interleave(boxes_per_img).float()
    return dict(img=img.to(device), cls=cls.to(device), bboxes=torch.cat((cxy, wh), 1).to(device), batch_idx=bi.to(device), prob=torch.ones(n, 1, device=device))


def build_model(cfg_name, device, seed=0):
    import yaml

    from cerberusdet_amd.models import CerberusDet

    cfg = yaml.safe_load(open(ROOT / "cerberusdet_amd" / "models" / "cfg" / cfg_name))
    torch.manual_seed(seed)
    model = CerberusDet(TASKS, NC, cfg=cfg, verbose=False)
    model.sequential_split(cfg["cerber"], "cpu")
    model.hyp = HYP
    return model.to(device).train(), cfg


def _cpu_baseline_worker(cfg, q):
    """Child process (never touches the GPU): the CPU oracle (pure torch fp32 restatement of the reference's graph / loss / NMS,
    pinned against the real reference by tests/golden) timed on the host cores -- BASELINE.md section 4:
      value  one task pass fwd + loss + bwd of the headline model (YOLOv8x 2-task) at bs 1 @640 -- the metric's workload, bounded;
      extra  BASELINE.md section 4's workloads with its protocol (3 warm-up + 10 timed iterations, median): (i) BASELINE config 1: YOLOv8n
             1-task bs 2 @640 fwd + loss + bwd + SGD-nesterov step, (ii) YOLOv8x 2-task all-heads eval forward bs 1 and bs 8,
             (iii) NMS (inference settings) on the section-8d prediction tensor, 128 images."""
    import yaml

    from oracle import graph as og
    from oracle import loss as ol
    from oracle import nms as on

    n_thr = max(1, min(len(os.sched_getaffinity(0)), 64))
    torch.set_num_threads(n_thr)
    try:
        cpu = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except Exception:
        cpu = "unknown"
    g = og.build_graph(cfg, TASKS, NC)
    og.apply_cerber_schedule(g, cfg["cerber"])
    w = og.init_weights(g, seed=0)
    def task_pass(ti, t):
        wt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in w.items()}
        b = synth_batch(0, ti, 0, 1, NC[ti], 640, "cpu")
        x = b["img"].float() / 255
        feats = og.forward(g, wt, x, t, training=True)
        scalar, _ = ol.detection_loss(feats, b, NC[ti], dict(box=7.5, cls=0.5, dfl=1.5))
        scalar.backward()

    task_pass(0, TASKS[0])  # untimed: first touch of the weights, thread-pool start-up
    t0 = time.perf_counter()
    n_img = 0
    for rep in range(2):  # the metric's iteration at batch 1: one pass per task, twice (~10-15 s on 64 cores)
        for ti, t in enumerate(TASKS):
            task_pass(ti, t)
            n_img += 1
    dt = time.perf_counter() - t0
    extra = {}

    def timed(fn, warm=3, n=10):
        """BASELINE.md section 4 protocol: 3 warm-up + 10 timed iterations, median (seconds)."""
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t1 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t1)
        return sorted(ts)[len(ts) // 2]

    try:
        # (ii) all-heads eval forward, bs 1 and bs 8
        with torch.no_grad():
            for bsx in (1, 8):
                xb = synth_batch(0, 0, 1, bsx, NC[0], 640, "cpu")["img"].float() / 255
                m = timed(lambda: og.forward(g, w, xb, None, training=False))
                extra[f"v8x_2task_allheads_forward_bs{bsx}"] = {"ms": round(m * 1e3, 1), "images_per_sec": round(bsx / m, 3)}
        # (i) config 1: YOLOv8n 1-task bs 2 @640, forward + loss + backward + SGD step
        cfgn = yaml.safe_load(open(ROOT / "cerberusdet_amd" / "models" / "cfg" / "v8n.yaml"))
        gn = og.build_graph(cfgn, ["voc"], [20])
        wn = og.init_weights(gn, seed=0)
        wtn = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in wn.items()}
        opt = torch.optim.SGD([v for v in wtn.values() if isinstance(v, torch.Tensor) and v.requires_grad], lr=0.00309, momentum=0.952, nesterov=True)
        bn = synth_batch(0, 0, 2, 2, 20, 640, "cpu")
        xn = bn["img"].float() / 255

        def step_n():
            opt.zero_grad()
            f = og.forward(gn, wtn, xn, "voc", training=True)
            sc, _ = ol.detection_loss(f, bn, 20, dict(box=7.5, cls=0.5, dfl=1.5))
            sc.backward()
            opt.step()

        m = timed(step_n)
        extra["config1_v8n_1task_bs2_train_step"] = {"ms": round(m * 1e3, 1), "images_per_sec": round(2 / m, 2)}
        # (iii) NMS (inference settings) on the section-8d prediction tensor, batch 128
        y = nms_inputs(128, 20, 8400, dtype=torch.float32).numpy()
        m = timed(lambda: on.non_max_suppression(y, conf_thres=0.25, iou_thres=0.45, max_det=300))
        extra["nms_infer_settings_128_images"] = {"ms": round(m * 1e3, 1)}
        extra["protocol"] = "BASELINE.md section 4: 3 warm-up + 10 timed iterations, median, fp32, all host cores"
    except Exception as e:  # the headline sample above stands on its own
        extra["error"] = repr(e)
    q.put(dict(value=round(n_img / dt, 4), unit="images/sec", cores=n_thr, cpu_model=cpu, kind="port",
               sample=f"{n_img} task passes (fwd + loss + bwd per task, no optimizer; both tasks, twice, after one untimed pass) of the YOLOv8x 2-task model @640 at batch 1, CPU oracle (torch fp32), {dt:.1f} s; extra = BASELINE.md section 4 workloads, 3 warm-up + 10 timed iterations each, median",
               extra=extra))


def cpu_baseline(cfg, timeout_s=300):
    """The reference's device='cpu' path cannot travel; time the parity-pinned CPU oracle (oracle/) on the host cores on a
    bounded sample of the same workload, in a child process with a hard time limit."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_baseline_worker, args=(cfg, q))
    p.start()
    try:
        res = q.get(timeout=timeout_s)
    except Exception:
        res = dict(value=None, unit="images/sec", cores=len(os.sched_getaffinity(0)), kind="port",
                   sample=f"CPU oracle did not finish one bs-1 task pass within {timeout_s} s")
    p.join(5)
    if p.is_alive():
        p.kill()
    return res


def kernel_breakdown(trainer, batches, n_max):
    """Replay one iteration with a HIP event pair around every C-ABI call (on the launch stream = torch's current stream) and
    attribute time + algorithmic conv FLOPs per entry point."""
    import ctypes as C

    from cerberusdet_amd import _lib as L

    lib = L.load()
    rec = []

    def instrument(calls):
        out = []
        for fn, args in calls:
            def wrapped(*a, _fn=fn, _args=args):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*a)
                e1.record()
                rec.append((_fn.__name__ if hasattr(_fn, "__name__") else str(_fn), _args, e0, e1))
                return rc
            wrapped.__name__ = getattr(fn, "__name__", "fn")
            out.append((wrapped, args))
        return out

    plans = [trainer.model.get_plan(t, batches[t]["img"].shape, batches[t]["img"].dtype, training=True) for t in TASKS]
    saved = [(p.fwd, p.bwd_groups) for p in plans]
    for p in plans:
        p.fwd = instrument(p.fwd)
        p.bwd_groups = [(i, instrument(c)) for i, c in p.bwd_groups]
    # the event pairs need the kernels alone on the GPU: replay with the sequential schedule (in the timed region the two task passes
    # share the GPU on two streams, where a stream-side event pair also spans the other stream's interleaved work)
    ts, trainer.task_streams = trainer.task_streams, False
    red, trainer.reducer.enabled = trainer.reducer.enabled, False  # rank 0 replays alone: no collectives in the replay
    try:
        trainer.train_step(batches, n_max=n_max)
        torch.cuda.synchronize()
    finally:
        trainer.task_streams, trainer.reducer.enabled = ts, red
    for p, (f, b) in zip(plans, saved):
        p.fwd, p.bwd_groups = f, b
    agg = {}
    for name, args, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        flops = 0.0
        key = name
        if name in ("cdet_conv2d", "cdet_conv2d_wgrad", "cdet_conv2d_wgrad_grouped", "cdet_conv2d_tiled", "cdet_conv2d_tiled_bn", "cdet_conv2d_tiled_dgrad",
                    "cdet_conv2d_s2_tiled", "cdet_conv2d_s2_tiled_bn", "cdet_conv2d_s2_tiled_dgrad"):
            d = args[0]._obj  # ctypes.byref(desc) keeps the descriptor
            if name == "cdet_conv2d_wgrad_grouped":  # args[2] layers of this geometry in one launch
                key = "cdet_conv2d_wgrad"
                flops = args[2] * 2.0 * d.N * d.Hd * d.Wd * d.Cd * d.Cs * d.kh * d.kw
            elif name in ("cdet_conv2d_s2_tiled", "cdet_conv2d_s2_tiled_bn"):  # stride-2 forward on the parity-plane kernel (csrc/conv_vt.hip)
                key = "cdet_conv2d_s2_tiled[fwd]"
                flops = 2.0 * d.N * d.Hd * d.Wd * d.Cd * d.Cs * d.kh * d.kw
            elif name == "cdet_conv2d_s2_tiled_dgrad":  # its data gradient: 4 class launches; FLOPs of the forward conv it differentiates
                key = "cdet_conv2d_s2_tiled[dgrad]"
                flops = 2.0 * d.N * d.Hs * d.Ws * d.Cs * d.Cd * d.kh * d.kw
            elif name.startswith("cdet_conv2d_tiled"):
                # the tap-resident kernel: its data gradient is a forward launch on the flipped operand (same FLOPs as the conv it differentiates)
                key = "cdet_conv2d_tiled[dgrad]" if name.endswith("dgrad") else "cdet_conv2d_tiled[fwd]"
                flops = 2.0 * d.N * d.Hd * d.Wd * d.Cd * d.Cs * d.kh * d.kw
            elif name == "cdet_conv2d":
                M = d.N * d.Hd * d.Wd
                if d.mode == L.CONV_DGRAD:
                    key = "cdet_conv2d[dgrad]"
                    # algorithmic FLOPs of dgrad = those of the forward conv it differentiates
                    flops = 2.0 * d.N * d.Hs * d.Ws * d.Cs * d.Cd * d.kh * d.kw
                else:
                    key = "cdet_conv2d[fwd]"
                    flops = 2.0 * M * d.Cd * d.Cs * d.kh * d.kw
            else:
                flops = 2.0 * d.N * d.Hd * d.Wd * d.Cd * d.Cs * d.kh * d.kw
        a = agg.setdefault(key, dict(ms=0.0, n=0, flops=0.0))
        a["ms"] += ms
        a["n"] += 1
        a["flops"] += flops
    return agg


def nms_inputs(bs, nc, na, seed=7, dtype=torch.float16):
    """boxes cx,cy~U(0,640), w,h~U(10,110); 500 random anchors per image get one class score ~U(.25,.95),
    every other score ~U(0,.01)."""
    g = torch.Generator().manual_seed(seed)
    y = torch.empty(bs, 4 + nc, na)
    y[:, 0:2] = torch.rand(bs, 2, na, generator=g) * 640
    y[:, 2:4] = torch.rand(bs, 2, na, generator=g) * 100 + 10
    y[:, 4:] = torch.rand(bs, nc, na, generator=g) * 0.01
    for b in range(bs):
        idx = torch.randperm(na, generator=g)[:500]
        cls = torch.randint(0, nc, (500,), generator=g)
        y[b, 4 + cls, idx] = torch.rand(500, generator=g) * 0.7 + 0.25
    return y.to(dtype)


def north_star_forward(model, device, bs=32, imgsz=640, reps=40, dtype=torch.bfloat16):
    """BASELINE.json north_star: YOLOv8x 2-task all-heads FORWARD at batch 32 @640 (eval form: BN folded into the conv epilogue,
    decode included), bf16 storage / fp32 accumulate, HIP-event timed on the launch stream in this process. Algorithmic work:
    381.31 GFLOP per image (README.md:241 of the reference) -> fraction of the 2.5 PF/s dense bf16 MFMA peak.
    dtype = torch.float16 (`north_star_fwd_fp16`, round 6): the same protocol on the fp16 plans -- the reference's own inference dtype
    (cerberusdet_inference.py:34-40 `model.half()`), three mantissa bits more than bf16 at the same MFMA rate: the plan on which BOTH halves of the
    north star -- >= 40 % of the MFMA peak AND boxes within 1e-3 of the image scale of the full-precision forward -- can hold at once."""
    name = "bf16" if dtype == torch.bfloat16 else "fp16"
    model.eval()
    model.bfloat16() if dtype == torch.bfloat16 else model.half()
    x = torch.rand(bs, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(3)).to(dtype).to(device)
    with torch.no_grad():
        # steady state: after a few seconds of host-side work (the instrumented replay above) the GPU sits in a low clock state and
        # takes ~1 s of continuous load to ramp back up -- 3 warm-up forwards measured 16.2 ms where the same forward runs 14.0 ms
        # right after a training loop (tools/debug/ns_probe.py); warm up for >= 1 s like the training measurement does with its steps
        t_w = time.perf_counter()
        n_w = 0
        warm_s = float(os.environ.get("CDET_NS_WARM_S", "1.0"))
        while n_w < 5 or time.perf_counter() - t_w < warm_s:
            model(x)
            n_w += 1
            if n_w % 10 == 0:
                torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            model(x)  # the reference-like DEFAULT call: fresh output tensors per call (round 4: the output launches are re-pointed at a new
            #           output set, engine.Plan.fresh_outputs -- no device copy)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            model(x, zero_copy=True)  # outputs are views of the plan's buffers (what CerberusDetInference / val.run consume)
        e1.record()
        torch.cuda.synchronize()
        ms_zc = e0.elapsed_time(e1) / reps
        # what was timed is checked: the outputs are finite, the default call's fresh tensors carry the bits of the plan's own, and a checksum of
        # `y` goes into the line (the full-size parity of this very plan: tests/test_gpu_eval_fullsize.py)
        fresh = model(x)
        home = model(x, zero_copy=True)
        torch.cuda.synchronize()
        finite = all(bool(torch.isfinite(y).all()) and all(bool(torch.isfinite(f).all()) for f in maps) for y, maps in fresh.values())
        same = all(torch.equal(fresh[t][0], home[t][0]) and all(torch.equal(a, b) for a, b in zip(fresh[t][1], home[t][1])) for t in fresh)
        checksum = {t: round(float(y.double().sum()), 3) for t, (y, _) in fresh.items()}
        # ... and against the SAME forward at the reference's precision (model.full_precision(), cerberusdet_amd/precise.py: the reference's fp32 numbers
        # to ~1e-6, tests/test_gpu_full_precision.py) on the batch's first two images: what 16-bit storage costs on the timed workload
        vs_full = None
        if os.environ.get("CDET_BENCH_FULL_PRECISION", "1") != "0":
            keep = {t: (y[:2].float().clone(), [f[:2].float().clone() for f in maps]) for t, (y, maps) in fresh.items()}
            model.full_precision()
            full = model(x[:2].contiguous())
            torch.cuda.synchronize()
            px = max(float((keep[t][0][:, :4] - full[t][0][:, :4]).abs().max()) for t in full)
            pr = max(float((keep[t][0][:, 4:] - full[t][0][:, 4:]).abs().max()) for t in full)
            vs_full = {
                "images": 2, "boxes_max_abs_over_image_size": float("%.3e" % (px / imgsz)),
                "within_1e-3": bool(px / imgsz <= 1e-3 and pr <= 1e-3),  # north_star: boxes within 1e-3 rel (of the 640-pixel image scale), probabilities 1e-3 abs
                "head_maps_rel_l2_max": float("%.3e" % max(float((a - b).norm() / b.norm()) for t in full for a, b in zip(keep[t][1], full[t][1]))),
                "boxes_max_abs_px": float("%.3e" % max(float((keep[t][0][:, :4] - full[t][0][:, :4]).abs().max()) for t in full)),
                "class_prob_max_abs": float("%.3e" % max(float((keep[t][0][:, 4:] - full[t][0][:, 4:]).abs().max()) for t in full)),
                "note": "random-init weights after the timed training steps: the head maps are dominated by their biases (models/yolo.py bias_init)",
            }
            del full
            model.bfloat16() if dtype == torch.bfloat16 else model.half()  # (switching the compute dtype drops the cached plans, the full-precision one included)
    tf = bs * 381.31e9 * (imgsz / 640) ** 2 / (ms * 1e-3) / 1e12
    model.train()
    return {"ms": round(ms, 3), "tflops": round(tf, 1), "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "images_per_sec": round(bs / ms * 1e3, 1),
            "config": f"YOLOv8x 2-task all-heads forward + decode, eval form (BN folded), {name}, batch {bs} @{imgsz}; default model(x) call (fresh output tensors)",
            "ms_with_fresh_output_tensors": round(ms, 3), "ms_zero_copy": round(ms_zc, 3),
            "outputs_finite": finite, "fresh_equals_zero_copy": same, "y_checksum": checksum, f"{name}_vs_full_precision": vs_full,
            "gflop_per_image": 381.31, "timing": f"HIP events around {reps} back-to-back forwards on the launch stream after {n_w} warm-up forwards (>= 1 s)"}


def inference_section(model, device, bs=128, imgsz=640, reps=10, nms_reps=50):
    """BASELINE.json configs[4]: CerberusDetInference-style fp16 all-heads forward at batch 128 @640 + batched NMS latency."""
    from cerberusdet_amd.utils.general import non_max_suppression

    out = {}
    model.eval().half()
    x = torch.rand(bs, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(3)).half().to(device)
    with torch.no_grad():
        t_w = time.perf_counter()
        n_w = 0
        while n_w < 2 or time.perf_counter() - t_w < 1.0:  # clock ramp-up, see north_star_forward
            model(x, zero_copy=True)
            torch.cuda.synchronize()
            n_w += 1
        t0 = time.perf_counter()
        for _ in range(reps):
            model(x, zero_copy=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["infer_images_per_sec"] = round(bs / dt, 1)
    out["infer_ms_per_batch"] = round(dt * 1e3, 2)
    out["infer_config"] = f"YOLOv8x 2-task all-heads forward + decode, fp16 storage, batch {bs} @{imgsz}, BN folded in the epilogue"
    out["infer_tflops"] = round(bs * 381.31e9 * (imgsz / 640) ** 2 / dt / 1e12, 1)
    y = nms_inputs(bs, 20, 8400).to(device)
    for name, kw in (("nms_infer", dict(conf_thres=0.25, iou_thres=0.45, max_det=300)),
                     ("nms_val", dict(conf_thres=0.001, iou_thres=0.6, multi_label=True, max_det=300))):
        ts = []
        for i in range(nms_reps + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = non_max_suppression(y, **kw)
            torch.cuda.synchronize()
            if i >= 3:
                ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out[name] = {"p50_ms": round(ts[len(ts) // 2], 3), "p95_ms": round(ts[int(len(ts) * 0.95)], 3), "batch": bs,
                     "kept_per_image": round(sum(r.shape[0] for r in res) / bs, 1), "settings": {k: v for k, v in kw.items()}}
    out.update(predict_e2e_child(device))
    model.train()
    model.bfloat16()
    return out


def _predict_e2e_worker(dev_index, q):
    try:
        device = torch.device("cuda", dev_index)
        torch.cuda.set_device(device)
        model, _ = build_model("v8x_2task.yaml", device)
        q.put(predict_e2e(model, device))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put({"infer_e2e_error": "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]})


def predict_e2e_child(device, timeout_s=300):
    """The end-to-end inference measurement (host frames -> result dicts, synchronous and as a stream) in a CHILD process with a hard time
    limit: it is the one section that drives side streams, pinned staging buffers and host threads, and it is secondary -- whatever happens
    in it, the bench line with the training metric is printed."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_predict_e2e_worker, args=(device.index if device.index is not None else 0, q))
    p.start()
    try:
        res = q.get(timeout=timeout_s)
    except Exception:
        res = {"infer_e2e_error": f"the end-to-end inference section did not finish within {timeout_s} s"}
    p.join(10)
    if p.is_alive():
        p.kill()
    return res


def calibrate_detections(det, x0, original_shape=None):
    """Shift every head's class-logit bias until about 100 detections per image SURVIVE NMS and the cross-task merge on the batch x0 (random
    boxes overlap heavily, so many more anchors than that have to pass the confidence threshold). With the reference's bias_init (class logit
    bias -10, models/yolo.py:102-110) nothing of a random-weight model would pass conf 0.25."""
    bs = x0.shape[0]
    with torch.no_grad():
        want_cand, applied = 400.0, {t: 0.0 for t in det.model.heads}
        for _ in range(6):
            out = det.model(x0, zero_copy=True)
            for t, (y, _) in out.items():
                best = y[:, 4:].float().amax(1).clamp(1e-7, 1 - 1e-7)            # [bs, A] best class probability per anchor
                logit = torch.log(best / (1 - best))
                q = torch.quantile(logit.flatten().float().cpu()[::4 * max(bs // 32, 1)], 1 - min(want_cand, 4000.0) / logit.shape[1])
                shift = float(math.log(0.25 / 0.75) - q)
                for lvl in range(3):
                    det.model.get_head(t).cv3[lvl][2].bias += shift
                applied[t] += shift
            det.model.mark_weights_changed()
            n_res = sum(len(r) for r in det.predict(x0, original_shape=original_shape)) / bs
            if 80 <= n_res <= 130 or want_cand >= 4000:
                break
            want_cand *= min(max(100.0 / max(n_res, 1.0), 0.25), 4.0)
    return n_res


def predict_e2e(model, device, bs=32, reps=5):
    """End-to-end CerberusDetInference.predict throughput (reference cerberusdet_inference.py + cerberusdet_preprocessor.py): host
    uint8 BGR 720x1280 frames -> upload + GPU letterbox -> fp16 all-heads forward -> per-task batched NMS -> cross-task merge +
    scale_boxes -> one D2H copy -> list of dicts. Synthetic frames, random-init weights. With the reference's bias_init (class logit
    bias -10) nothing would survive conf 0.25 and NMS / merge / dict building would run at their floor, so the class-logit biases of
    every head are shifted (calibrated on these frames) until about 100 anchors per image and task pass the confidence threshold:
    NMS, the cross-task merge, the D2H copy and the result dicts then do the work of a trained detector's busy frame."""
    import copy

    from cerberusdet_amd.cerberusdet_inference import CerberusDetInference
    from cerberusdet_amd.cerberusdet_preprocessor import CerberusPreprocessor

    model = copy.deepcopy(model)
    det = CerberusDetInference(model, device=str(device), half=True, img_size=640)
    pre = CerberusPreprocessor(img_size=640, stride=det.stride, half=True, auto=False)
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8) for _ in range(bs)]
    with torch.no_grad():
        x0 = pre.preprocess(frames, device)
    calibrate_detections(det, x0, (720, 1280))
    stages = {"preprocess_ms": [], "predict_ms": []}
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = pre.preprocess(frames, device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = det.predict(x, original_shape=(720, 1280))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if i >= 2:
            stages["preprocess_ms"].append((t1 - t0) * 1e3)
            stages["predict_ms"].append((t2 - t1) * 1e3)
    pm, qm = float(np.median(stages["preprocess_ms"])), float(np.median(stages["predict_ms"]))
    # the same work as a stream of batches (CerberusDetInference.predict_stream, depth 2): upload + letterbox + forward + NMS + merge of
    # batch i + 1 are enqueued before the result dicts of batch i are built, so the host work hides under the GPU's
    n_stream = 24

    def feed():
        for _ in range(n_stream):
            yield pre.preprocess(frames, device), (720, 1280)

    import gc

    with torch.no_grad():
        list(det.predict_stream(((pre.preprocess(frames, device), (720, 1280)) for _ in range(6)), depth=2))  # warm (the pinned staging ring)
        torch.cuda.synchronize()
        # what a serving process does once after start-up: the long-lived objects (model, plans) leave the collector's generations, so that the
        # full collections the ~5000 result dicts per batch trigger do not walk them (one 65 ms pause per ~25 batches otherwise)
        gc.collect()
        gc.freeze()
        t0 = time.perf_counter()
        n_out = sum(len(r) for r in det.predict_stream(feed(), depth=2))
        torch.cuda.synchronize()
        stream_ms = (time.perf_counter() - t0) * 1e3 / n_stream
        assert n_out == n_stream * bs
        # BASELINE config 5 end to end: the same stream at batch 128 (fp16, batched NMS, cross-task merge, result dicts)
        frames128, n128 = frames * (128 // bs), 8
        list(det.predict_stream(((pre.preprocess(frames128, device), (720, 1280)) for _ in range(4)), depth=2))
        torch.cuda.synchronize()
        gc.collect()
        t0 = time.perf_counter()
        n_out = sum(len(r) for r in det.predict_stream(((pre.preprocess(frames128, device), (720, 1280)) for _ in range(n128)), depth=2))
        torch.cuda.synchronize()
        stream128_ms = (time.perf_counter() - t0) * 1e3 / n128
        assert n_out == n128 * len(frames128)
        gc.unfreeze()
    return {"infer_e2e_images_per_sec": round(bs / ((pm + qm) * 1e-3), 1),
            "infer_e2e_pipelined_images_per_sec": round(bs / (stream_ms * 1e-3), 1),
            "infer_e2e_pipelined_b128_images_per_sec": round(len(frames128) / (stream128_ms * 1e-3), 1),
            "infer_e2e": {"batch": bs, "frame": "720x1280 BGR uint8 (host memory)", "preprocess_ms": round(pm, 2), "predict_ms": round(qm, 2),
                          "results_per_image": round(sum(len(r) for r in res) / bs, 1), "pipelined_ms_per_batch": round(stream_ms, 2), "pipelined_b128_ms_per_batch": round(stream128_ms, 2),
                          "pipelined": f"predict_stream over {n_stream} batches, 2 in flight: preprocess + predict of batch i + 1 enqueued before the dicts of batch i are built",
                          "note": "preprocess includes the PCIe upload of the raw frames; predict = forward + NMS + merge + D2H + dict build; class-logit biases shifted so that ~100 anchors per image and task pass conf 0.25 (random weights otherwise yield no detection)"}}


def dry_comm(args, virtual_ranks=3, batch=2, imgsz=128):
    """8-GPU readiness that one GPU can prove: the host ORDER in which every rank enqueues its collectives. One communicator serialises
    them in enqueue order, so two ranks that enqueue different sequences deadlock (or reduce the wrong tensors into each other). Here
    torch.distributed.all_reduce is replaced by a recorder -- (bytes, stream, position); nothing is sent -- and the trainer runs a few
    iterations once per VIRTUAL rank (its own rank id, its own synthetic shard) for the 2-task and the 3-task model, with SyncBatchNorm
    and gradient reduction on, task streams on, and a --skip-batches pattern (all tasks / first only / all / last only). The recorded
    sequences must be identical on all virtual ranks. Reference: train.py:140-143 (SyncBatchNorm), 182-184 (DDP), trainers/averaging.py:144-163."""
    import yaml

    from cerberusdet_amd.models import CerberusDet
    from cerberusdet_amd.trainers import Averaging

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29513")
    os.environ["CDET_REDUCE_ALWAYS"] = "1"
    os.environ["CDET_SYNCBN_PEER"] = "0"  # record the process-group form: every SyncBatchNorm exchange shows up as a collective (the peer-write form,
    #                                       peer_exchange.py, replaces exactly the non-gradient ones: `collectives_with_peer_exchange` below)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)  # a real (1-rank) group: the plans compile their SyncBN form
    real = dist.all_reduce
    log, streams = [], {}

    class _Done:
        def wait(self):
            return True

    def recorder(t, op=None, group=None, async_op=False):
        sid = streams.setdefault(torch.cuda.current_stream().cuda_stream, len(streams))
        log.append((t.numel() * t.element_size(), sid, bool(async_op)))  # async_op: a gradient bucket (trainers/averaging.GradReducer)
        return _Done() if async_op else None

    out = {"mode": "dry-comm", "virtual_ranks": virtual_ranks, "batch": batch, "imgsz": imgsz, "plans": {}}
    ok = True
    try:
        dist.all_reduce = recorder
        for cfg_name, tasks, ncs in (("v8x_2task.yaml", TASKS, NC), ("v8x_3task.yaml", TASKS + ["objects365_tableware"], NC + [12])):
            cfg = yaml.safe_load(open(ROOT / "cerberusdet_amd" / "models" / "cfg" / cfg_name))
            patterns = [list(tasks), tasks[:1], list(tasks), tasks[-1:]]
            seqs = []
            for vr in range(virtual_ranks):
                torch.manual_seed(0)
                model = CerberusDet(tasks, ncs, cfg=cfg, verbose=False)
                model.sequential_split(cfg["cerber"], "cpu")
                hyp = dict(HYP, box=[7.5] * len(tasks), cls=[0.5] * len(tasks), dfl=[1.5] * len(tasks))
                model.hyp = hyp
                model = model.to(device).train()
                tr = Averaging(device, model, hyp, tasks, epochs=100, nb=1000, rank=vr, world_size=virtual_ranks, sync_bn=True)
                tr.trace_cb = lambda kind, key, task: log.append(("hook", str(key), task))  # host position of every completed backward unit
                per_step = []
                for i, active in enumerate(patterns):
                    log.clear()
                    tr.train_step({t: synth_batch(vr, tasks.index(t), i, batch, ncs[tasks.index(t)], imgsz, device) for t in active}, n_max=8)
                    torch.cuda.synchronize()
                    per_step.append(list(log))
                seqs.append(per_step)
                del tr, model
            same = all(s_ == seqs[0] for s_ in seqs[1:])
            ok = ok and same

            def overlap(st):
                """Gradient bytes whose all-reduce is enqueued BEFORE the last backward launch of the iteration: everything in front of the last
                unit hook (that hook fires behind the final launches of the last pass -- the stem's row; what it sends can overlap nothing)."""
                last = max(i for i, e in enumerate(st) if e[0] == "hook")
                grad = [(i, e[0]) for i, e in enumerate(st) if e[0] != "hook" and e[2]]
                tot = sum(b for _, b in grad)
                return sum(b for i, b in grad if i < last), tot

            steps = []
            for a, st in zip(patterns, seqs[0]):
                coll = [e for e in st if e[0] != "hook"]
                early, tot = overlap(st)
                steps.append({"active_tasks": a, "collectives": len(coll), "bytes": sum(b for b, _, _ in coll),
                              "streams_used": len({s_ for _, s_, _ in coll}),
                              "syncbn_exchanges": sum(1 for _, _, g_ in coll if not g_),
                              "collectives_with_peer_exchange": sum(1 for _, _, g_ in coll if g_),
                              "gradient_bytes": tot, "gradient_bytes_enqueued_before_last_backward_launch": early,
                              "gradient_overlap_frac": round(early / max(tot, 1), 4)})
            out["plans"][cfg_name] = {"identical_on_all_ranks": same, "steps": steps}
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()
    out["identical_on_all_ranks"] = ok
    return out


def stub_ranks(args, rank, world, emit):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    mode = os.environ["CDET_BENCH_STUB"]
    if mode == "fail" and rank == world - 1:
        sys.exit(3)  # a rank that dies: the launcher's exit code has to reach the caller of `bench.py --gpus N`

    def step():
        time.sleep(0.002)
        t = torch.ones(4) * (rank + 1)
        dist.all_reduce(t)  # the stand-in's one exchange
        return float(t[0])

    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s_ = step()
    dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit({"metric": "images/sec train @640 YOLOv8x 2-task", "stub": True, "value": round(args.batch * len(TASKS) * world * args.steps / float(tt), 2),
              "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tt) / args.steps * 1e3, 3),
              "scaling": "weak", "sum_of_ranks": s_, "argv": sys.argv[1:], "master_port": os.environ.get("MASTER_PORT")})
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per task per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--cfg", default="v8x_2task.yaml")
    ap.add_argument("--dry-comm", action="store_true", help="record (do not execute) every collective of a few iterations for several virtual ranks and check that "
                    "all ranks enqueue the same sequence (2- and 3-task plans, SyncBatchNorm, --skip-batches pattern); prints one JSON line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="SyncBatchNorm across ranks (BASELINE.json configs[2]); default: per-GPU statistics")
    ap.add_argument("--no-infer", action="store_true", help="skip the inference + NMS section (secondary part of the metric)")
    ap.add_argument("--host-batches", action="store_true", help="secondary measurement: the batches live in pinned HOST memory and are uploaded inside the "
                    "timed region every step (the PCIe-inclusive rate; `value` proper keeps its inputs resident in HBM)")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON: everything native libraries print there while the run is on (RCCL's version banner at
    # communicator creation, for one) goes to stderr instead; the descriptor is restored for the final print
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    if args.dry_comm:
        emit(dry_comm(args))
        return

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, RCCL) before anything in this
        # process touches the GPU, relay rank 0's JSON line and the exit code (never re-exec a process that has initialised HIP)
        import socket
        import subprocess

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        os.dup2(real_stdout, 1)  # the children inherit the real stdout (each of them keeps it clean the same way)
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if os.environ.get("CDET_BENCH_STUB"):
        # tests/test_distributed_cpu.py: the launcher path without a GPU -- the ranks rendezvous over gloo and run the SAME timing protocol
        # (warm-up, barrier, K steps, barrier, MAX over ranks, rank 0 prints the one JSON line) around a stand-in step. Not a measurement.
        return stub_ranks(args, rank, world, emit)
    # CDET_BENCH_ONE_GPU=1 (tests only: tests/test_gpu_distributed.py): every rank on cuda:0 and the process group on gloo -- RCCL refuses two ranks on
    # one device -- so that the WHOLE N > 1 path of this file (rendezvous, weight broadcast, sharded batches, reducer, comm timer, MAX over ranks,
    # rank-0 line) runs on a one-GPU box. Not a measurement.
    one_gpu = os.environ.get("CDET_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("CDET_REDUCE_ALWAYS") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    from cerberusdet_amd.trainers import Averaging

    model, cfg = build_model(args.cfg, device)
    if use_dist:  # every rank starts from rank 0's weights (what DDP's constructor broadcast does in the reference, train.py:184)
        for t in list(model.state_dict().values()):
            dist.broadcast(t, src=0)
    trainer = Averaging(device, model, HYP, TASKS, epochs=100, nb=1000, rank=rank if use_dist else -1, world_size=world, sync_bn=args.sync_bn)
    n_iter = args.warmup + args.steps
    n_distinct = min(n_iter, 4)  # a few distinct synthetic batches, resident in HBM, cycled
    data = [{t: synth_batch(rank, ti, i, args.batch, NC[ti], args.imgsz, device) for ti, t in enumerate(TASKS)} for i in range(n_distinct)]
    n_max = 8
    if args.host_batches:  # what a DataLoader with pin_memory=True hands over (the reference's loaders, data/dataloaders.py:36)
        host = [{t: {k: v.cpu().pin_memory() for k, v in b.items()} for t, b in d.items()} for d in data]
        data = None

        class _Up:  # upload on access, so the copy sits inside the timed region of the step that uses it
            def __getitem__(self, i):
                return {t: {k: v.to(device, non_blocking=True) for k, v in b.items()} for t, b in host[i].items()}

        data = _Up()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    items = None
    # CDET_DEFER_TAIL=1 (A/B timing; measured 0.3 ms SLOWER, profiles/r05_optimizer_tail.txt, hence off): the optimizer's update of the unshared
    # blocks (necks + heads) on a side stream under the next iteration's trunk. Every update stays inside the timed region either way: sync() =
    # barrier + device-wide synchronize.
    defer = os.environ.get("CDET_DEFER_TAIL", "0") == "1"
    for i in range(args.warmup):
        items = trainer.train_step(data[i % n_distinct], n_max=n_max, defer_tail=defer)
    sync()
    comm_timer = None
    if use_dist:  # where the compute streams wait for communication (event pairs; trainers/averaging.py::CommTimer) -- read after the timed region
        from cerberusdet_amd.trainers.averaging import CommTimer

        comm_timer = model._comm_timer = CommTimer()
    t0 = time.perf_counter()
    for i in range(args.steps):
        items = trainer.train_step(data[(args.warmup + i) % n_distinct], n_max=n_max, defer_tail=defer)
    sync()
    trainer.join_tail()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    comm_exposed = None
    if comm_timer is not None:
        c = comm_timer.collect()
        model._comm_timer = None
        n_sp = int(c.pop("n_spans", 0))
        comm_exposed = {k: round(v / args.steps, 3) for k, v in c.items()}
        comm_exposed["total"] = round(sum(c.values()) / args.steps, 3)
        comm_exposed["spans_per_step"] = round(n_sp / args.steps, 1)
        comm_exposed["what"] = ("ms per iteration (this rank) a compute stream spent waiting for communication: grad_wait = the join on the gradient all-reduce "
                                "handles in front of the optimizer step, syncbn = the SyncBatchNorm statistics collectives (on their layer's dependent chain)")
    loss_items = {t: [round(float(v), 5) for v in items[t].tolist()] for t in TASKS}
    finite = all(np.isfinite(v).all() for v in loss_items.values())

    agg_all = None
    if not args.no_breakdown and args.sync_bn and world > 1:
        # SyncBatchNorm's per-layer collectives sit inside the launch lists: EVERY rank replays the instrumented iteration (the gradient
        # reducer stays off in the replay), rank 0 reports
        agg_all = kernel_breakdown(trainer, data[0], n_max)
    if rank == 0:
        imgs_per_step = args.batch * len(TASKS) * world
        value = imgs_per_step * args.steps / dt
        ms_per_step = dt / args.steps * 1e3
        step_tflop = 3 * GFLOP_FWD_PER_IMG_TASK * args.batch * len(TASKS) / 1e3 * (args.imgsz / 640) ** 2  # per GPU, fwd+bwd = 3x fwd
        out = {
            "metric": "images/sec train @640 YOLOv8x 2-task", "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"YOLOv8x 2-task (VOC nc20 + O365-animals nc19) training iteration, batch {args.batch}/task/GPU @{args.imgsz}, "
                                   "fwd+loss+bwd per task, clip+SGD-nesterov+EMA, bf16 storage / fp32 accumulate",
                       "cfg": args.cfg, "global_batch": imgs_per_step, "parallelism": f"dp{world}",
                                  **({"inputs": "pinned host memory, uploaded inside every timed step (PCIe-inclusive secondary measurement)"} if args.host_batches else {})},
            "step_tflop_per_gpu": round(step_tflop, 2), "achieved_tflops_per_gpu": round(step_tflop / (ms_per_step / 1e3), 1),
            "loss_items": loss_items, "loss_finite": bool(finite),
        }
        if comm_exposed is not None:
            out["comm_exposed_ms"] = comm_exposed
        try:
            from cerberusdet_amd import _lib as _L

            out["switches"] = _L.active_switches()  # every library switch off its default + the build flavour ("" = the product configuration)
            # ... and every CDET_* variable of the process environment: the host side reads its own (plan forms, schedules) when a plan is compiled
            out["switches_env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("CDET_")}
            st = trainer.scaler_state()
            out["grad_scaler"] = {"scale": st["scale"], "skipped_steps": st["skipped_steps"]}
        except Exception as e:  # noqa: BLE001
            out["switches"] = f"unavailable: {e}"
        if not args.no_breakdown:
            agg = agg_all if agg_all is not None else kernel_breakdown(trainer, data[0], n_max)
            tot = sum(a["ms"] for a in agg.values())
            out["kernel_ms"] = {k: round(a["ms"], 3) for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
            out["kernel_ms_total"] = round(tot, 3)
            # the dominant GPU kernel: the two tiled entry points (forward and stride-1 data gradient) launch the same
            # conv_halo_kernel, so they are one row here
            kern_of = {"cdet_conv2d_tiled[fwd]": "conv_halo_kernel", "cdet_conv2d_tiled[dgrad]": "conv_halo_kernel"}
            merged = {}
            for k, v in agg.items():
                if v["flops"] > 0:
                    m = merged.setdefault(kern_of.get(k, k), {"ms": 0.0, "flops": 0.0, "n": 0, "entries": []})
                    m["ms"] += v["ms"]
                    m["flops"] += v["flops"]
                    m["n"] += v["n"]
                    m["entries"].append(k)
            dom = max(merged, key=lambda k: merged[k]["ms"])
            a = merged[dom]
            ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
            out["roofline"] = {"kernel": dom, "entry_points": a["entries"], "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None, "launches": a["n"],
                               "avg_launch_ms": round(a["ms"] / a["n"], 4), "algorithmic_gflop_per_launch": round(a["flops"] / a["n"] / 1e9, 3),
                               "measured": "HIP events around every call of a sequential replay of one iteration (task streams off)"}
            # HBM traffic per launch of that entry point from the committed rocprofv3 PMC passes of this same command
            # (tools/pmc_traffic.py: FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; separate passes) -- null when absent
            tf = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))
            if tf:
                kern = json.load(open(tf[-1]))["kernels"]
                fam = [f"conv_igemm_{v}_kernel<0, {t}" for v in ("pipe", "glds") for t in ("2, 2", "3, 1", "4, 2")]
                main, extra = {"conv_halo_kernel": (("conv_halo_kernel", "conv_pair_kernel", "conv_pp_kernel"), ()),  # (the entry point's 1x1 layers with K >= 640 run the pair tile, its single-round 3x3 / 160-cout layers the 8-wave ping-pong form)
                               "cdet_conv2d_wgrad": (("wgrad_halo_kernel", "conv_wgrad_pipe_kernel", "conv_wgrad_kernel"), ("wgrad_reduce",)),
                               "cdet_conv2d[fwd]": (tuple(f"{f}, 0," for f in fam), ()),
                               "cdet_conv2d[dgrad]": (tuple(f"{f}, {m}," for f in fam for m in (1, 2)), ())}[dom]
                sel = [v for k, v in kern.items() if k.startswith(main + extra)]
                n0 = sum(v["launches"] for k, v in kern.items() if k.startswith(main))
                if sel and n0:
                    out["roofline"]["traffic"] = round(sum(v["fetch_bytes_total"] + v["write_bytes_total"] for v in sel) / n0)
                    out["roofline"]["traffic_unit"] = "bytes/launch"
                    out["roofline"]["traffic_source"] = tf[-1].name
            out["mfma_kernels"] = {k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1), "ms": round(v["ms"], 2), "launches": v["n"]}
                                   for k, v in agg.items() if v["flops"] > 0}
        if not args.no_infer and world == 1:
            out["north_star_fwd"] = north_star_forward(model, device, bs=args.batch, imgsz=args.imgsz)
            out["north_star_fwd_fp16"] = north_star_forward(model, device, bs=args.batch, imgsz=args.imgsz, dtype=torch.float16)
            model.bfloat16()
            out["inference"] = inference_section(model, device)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only: the other ranks would just wait at the barrier
            out["cpu_baseline"] = cpu_baseline(cfg)
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
