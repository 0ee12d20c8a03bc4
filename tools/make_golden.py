#!/usr/bin/env python3
"""Generate golden fixtures by running the REAL reference (ai-forever/CerberusDet) on the CPU.

Runs only in the build container (hard-codes /root/reference, refuses to run elsewhere). Nothing of
the reference travels: fixtures hold plain arrays (inputs, weights keyed by the reference's state-dict
schema, outputs) + JSON known answers. Absent third-party modules are stubbed before import
(SURVEY.md section 8c); `torchvision.ops.nms` is stubbed with the documented greedy algorithm
(that boundary stays "parity unpinned").

    python tools/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import copy
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden"
sys.path.insert(0, str(OUT))
import synth  # noqa: E402  (tests/golden/synth.py: deterministic inputs shared with the tests)


def _install_stubs():
    os.environ.setdefault("HOME", "/tmp")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Logger:
        def __getattr__(self, _):
            return lambda *a, **k: None

    mod("loguru", logger=_Logger())
    mod("cv2", setNumThreads=lambda *_: None, __version__="0")
    mod("seaborn")
    mod("thop")
    mlflow = mod("mlflow")
    mod("mlflow.models")
    mod("mlflow.models.signature", infer_signature=lambda *a, **k: None)
    mod("mlflow.tracking", MlflowClient=object)
    mlflow.models = sys.modules["mlflow.models"]
    mod("tensorboard")
    if "torch.utils.tensorboard" not in sys.modules:
        mod("torch.utils.tensorboard", SummaryWriter=object)
    for r in ("ray", "ray.air", "ray.tune", "ray.tune.experiment", "ray.tune.experiment.trial", "ray.tune.logger"):
        mod(r)

    def nms(boxes, scores, iou_threshold):
        # documented torchvision.ops.nms semantics (see oracle/nms.py header) -- parity unpinned.
        order = torch.argsort(scores, descending=True, stable=True)
        b = boxes[order]
        areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        n = b.shape[0]
        sup = torch.zeros(n, dtype=torch.bool)
        keep = []
        for i in range(n):
            if sup[i]:
                continue
            keep.append(i)
            if i + 1 < n:
                lt = torch.maximum(b[i, :2], b[i + 1:, :2])
                rb = torch.minimum(b[i, 2:], b[i + 1:, 2:])
                wh = (rb - lt).clamp(min=0)
                inter = wh[:, 0] * wh[:, 1]
                iou = inter / (areas[i] + areas[i + 1:] - inter)
                sup[i + 1:] |= iou > iou_threshold
        return order[torch.tensor(keep, dtype=torch.long)]

    tv = mod("torchvision")
    tv.ops = mod("torchvision.ops", nms=nms)


def _np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def tiny_cfg(three_tasks=False):
    import yaml

    name = "yolov8x_voc_obj365_animals_tableware.yaml" if three_tasks else "yolov8x_voc_obj365.yaml"
    cfg = yaml.safe_load(open(REF / "cerberusdet/models" / name))
    cfg["depth_multiple"] = synth.TINY_DEPTH
    cfg["width_multiple"] = synth.TINY_WIDTH  # channels 8/16/32/64/64 -> small fixtures
    return cfg


HYP = synth.HYP


def assign_det_weights(model, seed):
    """Every state-dict entry <- synth.det_tensor(seed, key, shape) (no weights need to be stored)."""
    sd = model.state_dict()
    for k, v in sd.items():
        v.copy_(torch.from_numpy(synth.det_tensor(seed, k, v.shape)))
    model.load_state_dict(sd)


def make_batch(bs, n_per_img, nc, seed, empty_images=()):
    return {k: torch.from_numpy(v) for k, v in synth.make_batch(bs, n_per_img, nc, seed, empty_images).items()}


def gen_graph_known_answers():
    """Block numbering / execution plans / FLOPs+params of the shipped YAMLs (SURVEY.md section 8a a7,a8; README)."""
    from cerberusdet.models.cerberus import CerberusDet

    out = {}
    for name, tasks, nc in [
        ("yolov8x_voc_obj365.yaml", ["voc", "objects365_animals"], [20, 19]),
        ("yolov8x_voc_obj365_animals_tableware.yaml", ["voc", "objects365_animals", "objects365_tableware"], [20, 19, 12]),
        ("yolov8x.yaml", ["voc"], [20]),
    ]:
        import yaml

        cfg = yaml.safe_load(open(REF / "cerberusdet/models" / name))
        m = CerberusDet(task_ids=tasks, nc=nc, cfg=copy.deepcopy(cfg), ch=3, verbose=False)
        if cfg.get("cerber"):
            m.sequential_split(copy.deepcopy(cfg["cerber"]), "cpu")
        entry = dict(
            n_blocks=len(m.blocks), heads={k: int(v) for k, v in m.heads.items()},
            plans={t: [int(i) for i in m.execution_plan(t)[0]] for t in tasks},
            plan_all=[int(i) for i in m.execution_plan(tasks)[0]],
            branching_points=sorted(int(i) for i in m.branching_points),
            stride=[float(s) for s in m.stride],
            n_params=int(sum(p.numel() for p in m.parameters())),
            n_state_keys=len(m.state_dict()),
            serving={str(c.index): list(c.serving_tasks.keys()) for c in m.controllers},
            block_types=[type(b).__name__ for b in m.blocks],
        )
        # conv FLOPs per image @640 = 2*MACs of every nn.Conv2d executed (hooks), per task path and all heads
        flops = {}
        for label, tids in [(t, t) for t in tasks] + [("all", None)]:
            total = [0]
            hooks = []

            def hook(mod, inp, outp):
                total[0] += 2 * outp.numel() * mod.in_channels // mod.groups * mod.kernel_size[0] * mod.kernel_size[1]

            for mod_ in m.modules():
                if isinstance(mod_, torch.nn.Conv2d):
                    hooks.append(mod_.register_forward_hook(hook))
            m.eval()
            with torch.no_grad():
                m(torch.zeros(1, 3, 640, 640), tids)
            for h in hooks:
                h.remove()
            flops[label] = int(total[0])
        entry["conv_flops_640"] = flops
        head = m.get_head(tasks[0])
        entry["cls_bias_init"] = [float(head.cv3[i][-1].bias[0]) for i in range(3)]
        entry["state_keys_sample"] = sorted(m.state_dict().keys())[:40]
        out[name] = entry
    return out


def gen_model_fixture(three_tasks, seed, bs, imgsz, fname):
    from cerberusdet.models.cerberus import CerberusDet

    cfg = tiny_cfg(three_tasks)
    tasks = ["voc", "objects365_animals", "objects365_tableware"][: 3 if three_tasks else 2]
    nc = [20, 19, 12][: len(tasks)]
    m = CerberusDet(task_ids=tasks, nc=nc, cfg=copy.deepcopy(cfg), ch=3, verbose=False)
    m.sequential_split(copy.deepcopy(cfg["cerber"]), "cpu")
    assign_det_weights(m, seed)  # after the split: clones get their own values -> wiring errors are visible

    x = torch.from_numpy(synth.det_image(seed, bs, imgsz))
    arrays = {}
    sd_shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    meta = dict(tasks=tasks, nc=nc, cfg=cfg, torch=torch.__version__, bs=bs, imgsz=imgsz, seed=seed,
                plans={t: [int(i) for i in m.execution_plan(t)[0]] for t in tasks},
                plan_all=[int(i) for i in m.execution_plan(tasks)[0]], heads={k: int(v) for k, v in m.heads.items()},
                stride=[float(s) for s in m.stride], state_shapes=sd_shapes)
    # train-mode forward per task (BN batch stats) + input-gradient and a few weight-gradients of sum(maps * cot)
    m.train()
    for t in tasks:
        sd0 = copy.deepcopy(m.state_dict())
        xg = x.clone().requires_grad_(True)
        feats = m(xg, t)
        r = [torch.from_numpy(synth.det_array(seed, f"cot/{t}/{i}", f.shape)) for i, f in enumerate(feats)]
        m.zero_grad()
        sum((f * ri).sum() for f, ri in zip(feats, r)).backward()
        for i, f in enumerate(feats):
            arrays[f"train/{t}/feat{i}"] = f.detach().numpy()
        arrays[f"train/{t}/dx"] = xg.grad.numpy()
        named = dict(m.named_parameters())
        gk = [k for k in named if named[k].grad is not None]
        pick = [gk[0], gk[len(gk) // 3], gk[len(gk) // 2], gk[-2], gk[-1]]
        pick += [k for k in gk if k.endswith("bn.weight")][:2] + [k for k in gk if k.endswith("bn.bias")][-2:]
        pick += [k for k in gk if ".m.0.cv1.conv.weight" in k][:2]
        for k in sorted(set(pick)):
            arrays[f"train/{t}/grad/{k}"] = named[k].grad.numpy().copy()
        meta.setdefault("grad_keys_with_grad", {})[t] = sorted(gk)
        # BN running stats after this one training forward (a few of them)
        sd1 = m.state_dict()
        changed = [k for k in sd1 if k.endswith(("running_mean", "running_var")) and not torch.equal(sd1[k], sd0[k])]
        meta.setdefault("bn_changed", {})[t] = len(changed)
        for k in changed[:4] + changed[-4:]:
            arrays[f"train/{t}/bn/{k}"] = sd1[k].numpy().copy()
        m.load_state_dict(sd0)
    # eval-mode (running stats), all heads: y + feats
    m.eval()
    with torch.no_grad():
        out = m(x)
    for t in tasks:
        y, feats = out[t]
        arrays[f"eval/{t}/y"] = y.numpy()
        for i, f in enumerate(feats):
            arrays[f"eval/{t}/feat{i}"] = f.numpy()
    # fused (conv+BN folded) eval forward
    mf = copy.deepcopy(m).fuse().eval()
    with torch.no_grad():
        outf = mf(x)
    for t in tasks:
        arrays[f"fused/{t}/y"] = outf[t][0].numpy()
    np.savez_compressed(OUT / fname, **arrays)
    json.dump(meta, open(OUT / (fname.replace(".npz", ".json")), "w"), indent=1)
    return m, tasks, nc, cfg


def gen_clone_fixture():
    """sequential_split deep-copies weights of the cloned blocks (cerberus.py:530): record (clone -> source)."""
    from cerberusdet.models.cerberus import CerberusDet

    out = {}
    for three in (False, True):
        cfg = tiny_cfg(three)
        tasks = ["voc", "objects365_animals", "objects365_tableware"][: 3 if three else 2]
        nc = [20, 19, 12][: len(tasks)]
        m = CerberusDet(task_ids=tasks, nc=nc, cfg=copy.deepcopy(cfg), ch=3, verbose=False)
        n0 = len(m.blocks)
        for i, b in enumerate(m.blocks):  # tag each original block through one of its tensors
            for p in b.parameters():
                p.data.fill_(float(i))
        m.sequential_split(copy.deepcopy(cfg["cerber"]), "cpu")
        src = {}
        for i in range(n0, len(m.blocks)):
            ps = list(m.blocks[i].parameters())
            src[str(i)] = int(ps[0].flatten()[0]) if ps else None
        inputs = {str(i): [list(x) if isinstance(x, tuple) else int(x) for x in getattr(b, "f", [])]
                  for i, b in enumerate(m.blocks) if i > 0}
        out["3task" if three else "2task"] = dict(n_blocks=len(m.blocks), clone_source=src, block_f=inputs)
    return out


def gen_loss_fixtures():
    from cerberusdet.utils.loss import Loss

    class _Head:
        def __init__(self, nc):
            self.nc, self.no, self.reg_max, self.stride = nc, nc + 64, 16, torch.tensor([8.0, 16.0, 32.0])

    class _Model(torch.nn.Module):
        def __init__(self, nc):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.hyp = HYP
            self.heads = {"t": 0}
            self._h = _Head(nc)

        def get_head(self, _):
            return self._h

    arrays, meta = {}, {}
    for name, (bs, imgsz, nc, npi, empty, seed, mode) in synth.LOSS_CASES.items():
        model = _Model(nc)
        crit = Loss(model, ["t"])
        batch = make_batch(bs, max(npi, 1), nc, seed, empty_images=empty if npi else tuple(range(bs)))
        feats = [torch.from_numpy(f).requires_grad_(True) for f in synth.synth_feats(seed, bs, imgsz, nc, mode)]
        loss, items = crit(feats, dict(batch, img=None), "t")
        loss.backward()
        # also capture the assigner outputs by re-running the reference's internals
        from cerberusdet.utils.tal import make_anchors

        with torch.no_grad():
            no = nc + 64
            pd, ps = torch.cat([f.view(bs, no, -1) for f in feats], 2).split((64, nc), 1)
            ps, pd = ps.permute(0, 2, 1).contiguous(), pd.permute(0, 2, 1).contiguous()
            ap, st = make_anchors(feats, crit.stride, 0.5)
            tg = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["prob"].view(-1, 1),
                            batch["bboxes"]), 1)
            sz = torch.tensor([imgsz, imgsz, imgsz, imgsz], dtype=torch.float32)
            tg = crit.preprocess(tg, bs, scale_tensor=sz)
            gl, gp, gb = tg.split((1, 1, 4), 2)
            mg = gb.sum(2, keepdim=True).gt_(0)
            pb = crit.bbox_decode(ap, pd)
            tl, tb, ts, fg, tgi = crit.assigner["t"](ps.sigmoid(), (pb * st).type(gb.dtype), ap * st, gl, gb, mg)
        p = f"{name}/"
        for i, f in enumerate(feats):
            arrays[p + f"dfeat{i}"] = f.grad.numpy()
        arrays[p + "loss"] = loss.detach().numpy()
        arrays[p + "items"] = items.numpy()
        arrays[p + "target_labels"] = tl.numpy().astype(np.int32)
        arrays[p + "target_bboxes"] = tb.numpy()
        arrays[p + "target_scores"] = ts.numpy()
        arrays[p + "fg_mask"] = fg.numpy().astype(np.uint8)
        arrays[p + "target_gt_idx"] = tgi.numpy().astype(np.int32)
        arrays[p + "padded_targets"] = tg.numpy()
        meta[name] = dict(bs=bs, imgsz=imgsz, nc=nc, gains=dict(box=7.5, cls=0.5, dfl=1.5), n_fg=int(fg.sum()))
    np.savez_compressed(OUT / "loss.npz", **arrays)
    json.dump(meta, open(OUT / "loss.json", "w"), indent=1)


def gen_nms_mask_fixture():
    """non_max_suppression with nm > 0 from the real reference. (Its `labels=` branch cannot be run: general.py:432 builds the label rows
    nc + nm + 5 columns wide -- a YOLOv5 leftover -- and the torch.cat with the nc + nm + 4 wide candidates raises.)"""
    import cerberusdet.utils.general as G
    from cerberusdet.utils.general import non_max_suppression

    G.time = types.SimpleNamespace(time=lambda: 0.0)
    arrays, meta = {}, {}
    for name, c in synth.NMS_MASK_CASES.items():
        y = synth.nms_mask_input(name)
        out = non_max_suppression(torch.from_numpy(y), **c["kw"])
        for i, o in enumerate(out):
            arrays[f"{name}/out{i}"] = o.numpy()
        meta[name] = dict(counts=[int(o.shape[0]) for o in out], cols=int(out[0].shape[1]))
    np.savez_compressed(OUT / "nms_masks.npz", **arrays)
    json.dump(meta, open(OUT / "nms_masks.json", "w"), indent=1)


def gen_nms_fixtures():
    import cerberusdet.utils.general as G
    from cerberusdet.utils.general import nms_between_tasks, non_max_suppression

    # the reference's wall-clock bail-out (general.py:417,477-479) silently truncates batches on a slow CPU;
    # freeze its clock so that every image is processed (the bail-out is deliberately not part of parity).
    G.time = types.SimpleNamespace(time=lambda: 0.0)

    arrays, meta = {}, {}
    for name, c in synth.NMS_CASES.items():
        y = synth.nms_case_input(name)
        out = non_max_suppression(torch.from_numpy(y), **c["kw"])
        for i, o in enumerate(out):
            arrays[f"{name}/out{i}"] = o.numpy()
        meta[name] = dict(kw=c["kw"], bs=c["bs"], counts=[int(o.shape[0]) for o in out])
    y = synth.ties_input()
    out = non_max_suppression(torch.from_numpy(y), conf_thres=0.25, iou_thres=1.0 / 3.0)
    arrays["ties/out0"] = out[0].numpy()
    meta["ties"] = dict(kw=dict(conf_thres=0.25, iou_thres=1.0 / 3.0), bs=1, counts=[int(out[0].shape[0])])

    # cross-task NMS + predict() post-processing (cerberusdet_inference.py:121-184) on synthetic per-task y
    from cerberusdet.cerberusdet_inference import CerberusDetInference

    ya, yb, names, shapes = synth.predict_inputs()
    inf = object.__new__(CerberusDetInference)
    inf.conf_thres, inf.iou_thres, inf.iou_thres_between_tasks = 0.25, 0.45, 0.8
    inf.names = names
    inf.categories_inds_map, inf.all_class_names = inf._get_categories_map(names)
    inf.model = lambda t: {"voc": (torch.from_numpy(ya), None), "objects365_animals": (torch.from_numpy(yb), None)}
    res = inf.predict(torch.zeros(3, 3, 640, 640), original_shape=list(shapes))
    meta["predict"] = dict(results=res, n_per_image=[len(r) for r in res])
    # direct nms_between_tasks case
    det = []
    for t, y in (("voc", ya), ("objects365_animals", yb)):
        o = non_max_suppression(torch.from_numpy(y[:1]), 0.25, 0.45)[0]
        o[:, 5] += 0 if t == "voc" else 20
        det.append(o)
    det = torch.cat(det, 0)
    arrays["between/in"] = det.numpy().copy()
    arrays["between/out"] = nms_between_tasks(det, inf.categories_inds_map, 0.8).numpy()
    np.savez_compressed(OUT / "nms.npz", **arrays)
    json.dump(meta, open(OUT / "nms.json", "w"), indent=1)


def gen_trainer_fixture(m, tasks, nc, cfg):
    """Two iterations of Averaging's inner loop + optimizer_step (trainers/averaging.py:142-223) on the tiny
    2-task model (weights = model_tiny2's: synth.det_tensor(seed=1, ...))."""
    from cerberusdet.trainers.averaging import Averaging, get_optimizer
    from cerberusdet.utils.loss import Loss
    from cerberusdet.utils.torch_utils import ModelEMA

    m = copy.deepcopy(m)
    m.hyp = HYP
    m.train()
    ema = ModelEMA(m)
    tr = object.__new__(Averaging)
    tr.optimizer = get_optimizer(m, HYP)
    tr.scaler = torch.amp.GradScaler("cuda", enabled=False)
    crit = Loss(m, tasks)
    num_branches = {idx: torch.tensor(float(max(len(c.serving_tasks), 1.0))) for idx, (c, b) in enumerate(m.control_blocks())}
    arrays, meta = {}, dict(tasks=tasks, nc=nc, hyp=HYP, iters=[], bs=2, imgsz=64, model_seed=1)
    bs, imgsz = 2, 64
    h0, h1, last = m.heads[tasks[0]], m.heads[tasks[1]], len(m.blocks) - 1
    watch = ["blocks.0.model.0.conv.weight", "blocks.0.model.0.bn.weight", "blocks.3.cv1.conv.weight", "blocks.3.cv1.bn.weight",
             f"blocks.{h0}.cv3.0.2.bias", f"blocks.{h1}.cv2.1.2.weight", "blocks.0.model.2.cv1.bn.running_mean",
             "blocks.0.model.2.cv1.bn.running_var", "blocks.0.model.2.cv1.bn.bias"]
    watch += [k for k in m.state_dict() if k.startswith(f"blocks.{last}.") and k.endswith("conv.weight")][:1]
    tr.optimizer.zero_grad()
    for it in range(2):
        info = {}
        for ti, t in enumerate(tasks):
            x = torch.from_numpy(synth.det_image(100 + 10 * it + ti, bs, imgsz))
            batch = make_batch(bs, 2, nc[ti], 200 + 10 * it + ti)
            out = m(x, t)
            loss, items = crit(out, dict(batch, img=x), t)
            loss.backward()
            arrays[f"it{it}/{t}/items"] = items.numpy()
            info[t] = float(loss)
        named = dict(m.named_parameters())
        gsq = sum(float((p.grad.double() ** 2).sum()) for p in named.values() if p.grad is not None)
        info["grad_norm"] = gsq ** 0.5
        tr.optimizer_step(m, ema, num_branches)
        sd, esd = m.state_dict(), ema.ema.state_dict()
        for k in watch:
            arrays[f"it{it}/w/{k}"] = sd[k].numpy().copy()
            arrays[f"it{it}/ema/{k}"] = esd[k].numpy().copy()
        meta["iters"].append(info)
    meta["watch"] = watch
    meta["num_branches"] = {str(k): float(v) for k, v in num_branches.items()}
    meta["param_group_sizes"] = [len(g["params"]) for g in tr.optimizer.param_groups]
    np.savez_compressed(OUT / "trainer.npz", **arrays)
    json.dump(meta, open(OUT / "trainer.json", "w"), indent=1)


def gen_train_wc_fixture():
    """Well-conditioned train-mode fixture (tests/golden/train_wc.npz): the tiny 2-task model with synth.det_tensor_wc weights at
    batch 8 @128. Part A: one forward + Loss + backward per task from the initial weights (head maps, loss items, EVERY parameter
    gradient). Part B: two iterations of Averaging's inner loop + optimizer_step + EMA (trainers/averaging.py:142-223): loss items,
    the total gradient norm, every parameter and a few running statistics / EMA entries after each iteration
    (gradients / weights as strided samples of <= ~1024 elements per tensor, synth.sample)."""
    from cerberusdet.models.cerberus import CerberusDet
    from cerberusdet.trainers.averaging import Averaging, get_optimizer
    from cerberusdet.utils.loss import Loss
    from cerberusdet.utils.torch_utils import ModelEMA

    W = synth.TRAIN_WC
    seed, bs, imgsz, nbox = W["seed"], W["bs"], W["imgsz"], W["boxes_per_img"]
    cfg = tiny_cfg(False)
    tasks, nc = ["voc", "objects365_animals"], [20, 19]
    m = CerberusDet(task_ids=tasks, nc=nc, cfg=copy.deepcopy(cfg), ch=3, verbose=False)
    m.sequential_split(copy.deepcopy(cfg["cerber"]), "cpu")
    sd = m.state_dict()
    for k, v in sd.items():
        v.copy_(torch.from_numpy(synth.det_tensor_wc(seed, k, v.shape)))
    m.load_state_dict(sd)
    m.hyp = HYP
    m.train()
    arrays, meta = {}, dict(tasks=tasks, nc=nc, cfg=cfg, hyp=HYP, torch=torch.__version__, iter_info=[], **W)
    crit = Loss(m, tasks)
    # ---- part A
    sd0 = copy.deepcopy(m.state_dict())
    for ti, t in enumerate(tasks):
        x = torch.from_numpy(synth.det_image(300 + ti, bs, imgsz))
        batch = make_batch(bs, nbox, nc[ti], 400 + ti)
        m.zero_grad()
        feats = m(x, t)
        loss, items = crit(feats, dict(batch, img=x), t)
        loss.backward()
        for i, f in enumerate(feats):
            arrays[f"A/{t}/feat{i}"] = synth.sample(f.detach().numpy(), 16384)
        arrays[f"A/{t}/items"] = items.numpy()
        meta.setdefault("A_loss", {})[t] = float(loss)
        for k, p in m.named_parameters():
            if p.grad is not None and float(p.grad.abs().max()) > 0:
                arrays[f"A/{t}/grad/{k}"] = synth.sample(p.grad.numpy())  # strided sample of <= ~1024 elements per tensor
        m.load_state_dict(sd0)  # BN running statistics back to the start
    m.zero_grad()
    # ---- part B
    ema = ModelEMA(m)
    tr = object.__new__(Averaging)
    tr.optimizer = get_optimizer(m, HYP)
    tr.scaler = torch.amp.GradScaler("cuda", enabled=False)
    num_branches = {idx: torch.tensor(float(max(len(c.serving_tasks), 1.0))) for idx, (c, b) in enumerate(m.control_blocks())}
    tr.optimizer.zero_grad()
    stat_keys = [k for k in m.state_dict() if k.endswith(("running_mean", "running_var"))]
    stat_keys = stat_keys[:4] + stat_keys[len(stat_keys) // 2: len(stat_keys) // 2 + 4] + stat_keys[-4:]
    for it in range(W["iters"]):
        info = {}
        for ti, t in enumerate(tasks):
            x = torch.from_numpy(synth.det_image(500 + 10 * it + ti, bs, imgsz))
            batch = make_batch(bs, nbox, nc[ti], 600 + 10 * it + ti)
            out = m(x, t)
            loss, items = crit(out, dict(batch, img=x), t)
            loss.backward()
            arrays[f"B/it{it}/{t}/items"] = items.numpy()
            info[t] = float(loss)
        gsq = sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None)
        info["grad_norm"] = gsq ** 0.5
        tr.optimizer_step(m, ema, num_branches)
        sd, esd = m.state_dict(), ema.ema.state_dict()
        for k, p in m.named_parameters():
            arrays[f"B/it{it}/w/{k}"] = synth.sample(sd[k].numpy())
        for k in stat_keys:
            arrays[f"B/it{it}/w/{k}"] = sd[k].numpy().copy()
        for k in list(dict(m.named_parameters()))[::12]:
            arrays[f"B/it{it}/ema/{k}"] = synth.sample(esd[k].numpy())
        meta["iter_info"].append(info)
    meta["stat_keys"] = stat_keys
    np.savez_compressed(OUT / "train_wc.npz", **arrays)
    json.dump(meta, open(OUT / "train_wc.json", "w"), indent=1)


def main():
    if not REF.exists():
        sys.exit("make_golden.py needs /root/reference (build container only)")
    OUT.mkdir(parents=True, exist_ok=True)
    _install_stubs()
    sys.path.insert(0, str(REF))
    torch.set_num_threads(8)
    torch.manual_seed(0)
    if sys.argv[1:] == ["nms_masks"]:  # only the mask-branch NMS fixture
        gen_nms_mask_fixture()
        print("nms_masks done", json.load(open(OUT / "nms_masks.json")))
        return
    if sys.argv[1:] == ["train_wc"]:  # only the well-conditioned train fixture (the others are untouched)
        gen_train_wc_fixture()
        print("train_wc done", (OUT / "train_wc.npz").stat().st_size // 1024, "KiB")
        return
    ka = gen_graph_known_answers()
    ka["clones"] = gen_clone_fixture()
    json.dump(ka, open(OUT / "graph_known_answers.json", "w"), indent=1)
    print("graph known answers done")
    m, tasks, nc, cfg = gen_model_fixture(False, 1, 2, 64, "model_tiny2.npz")
    print("model_tiny2 done")
    gen_model_fixture(True, 5, 1, 96, "model_tiny3.npz")
    print("model_tiny3 done")
    gen_loss_fixtures()
    print("loss done")
    gen_nms_fixtures()
    gen_nms_mask_fixture()
    print("nms done")
    gen_trainer_fixture(m, tasks, nc, cfg)
    print("trainer done")
    gen_train_wc_fixture()
    print("train_wc done")
    for f in sorted(OUT.iterdir()):
        print(f"{f.name:32s} {f.stat().st_size / 1024:9.1f} KiB")


if __name__ == "__main__":
    main()
