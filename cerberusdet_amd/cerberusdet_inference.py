"""`CerberusDetInference` with the reference's constructor / attributes / `predict` contract
(ai-forever/CerberusDet cerberusdet/cerberusdet_inference.py:19-54, 85-186).

Execution differs: the all-heads forward is one compiled gfx950 launch list (backbone + shared neck once, each branch once),
per-task NMS is ONE batched launch sequence per task, and the per-image combine / cross-task NMS / rescale runs on the few
hundred surviving rows after a single device->host copy (the reference does `.cpu()` + python lambdas per task per image).

Beyond the reference's synchronous `predict`: `predict_async` enqueues forward + NMS + merge + an asynchronous copy into pinned host
memory and returns a `PendingPrediction` whose `.result()` builds the reference's list of dicts; `predict_stream` keeps a few batches in
flight so that the host-side dict building of batch i runs under the GPU work of batch i + 1 (same results, same order).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Tuple, Union

import numpy as np
import torch

from .models import CerberusDet
from .utils.general import check_img_size, nms_between_tasks, non_max_suppression, scale_boxes


def checkpoint_dict(model: CerberusDet, names: Dict[str, List[str]] = None) -> dict:
    """Portable checkpoint: plain state dict in the reference's key schema + what is needed to rebuild the graph."""
    return dict(format="cerberusdet_amd/1", cfg=model.yaml, task_ids=list(model.heads.keys()),
                nc=[model.get_head(t).nc for t in model.heads], names=names or getattr(model, "names", None),
                fused=any(getattr(m, "fused", False) for m in model.modules()), state_dict=model.state_dict())


def save_checkpoint(path, model: CerberusDet, names: Dict[str, List[str]] = None):
    torch.save(checkpoint_dict(model, names), path)


def attempt_load(weights, map_location=None) -> CerberusDet:
    """Load a cerberusdet_amd checkpoint (reference models/experimental.py:99-139 unpickles whole nn.Modules by class path,
    which torch >= 2.6 refuses by default; this loader takes the state-dict schema instead)."""
    if isinstance(weights, CerberusDet):
        model = weights
    else:
        ck = torch.load(str(weights), map_location="cpu", weights_only=False)
        if not (isinstance(ck, dict) and ck.get("format", "").startswith("cerberusdet_amd/")):
            raise ValueError(f"{weights}: not a cerberusdet_amd checkpoint (see INTEGRATION.md for converting reference weights)")
        model = CerberusDet(ck["task_ids"], ck["nc"], cfg=ck["cfg"], verbose=False)
        if ck["cfg"].get("cerber"):
            model.sequential_split(ck["cfg"]["cerber"], "cpu")
        if ck.get("fused"):
            model.fuse()
        model.load_state_dict(ck["state_dict"])
        if ck.get("names"):
            model.names = ck["names"]
    if map_location is not None:
        model = model.to(map_location)
    return model.eval()


class PendingPrediction:
    """The device work of one `predict` call in flight: rows [bs, max_det, 6] and counts [bs] on their way into pinned host memory behind
    `event`. `.result()` waits for the event and builds the reference's output (cerberusdet_inference.py:150-186): per image a list of
    {"box": [x1, y1, x2, y2] ints, "score", "label" (global id), "label_name", "task"}."""

    def __init__(self, rows, cnt, event, class_names, task_of_label):
        self._rows, self._cnt, self._event, self._names, self._task_of = rows, cnt, event, class_names, task_of_label
        self._done = None

    def ready(self) -> bool:
        return self._done is not None or self._event.query()

    def result(self) -> List[List[Dict]]:
        if self._done is None:
            self._event.synchronize()
            rows, cnt, names, task_of = self._rows.numpy(), self._cnt.numpy().tolist(), self._names, self._task_of
            out = []
            for i, n in enumerate(cnt):
                r = rows[i, :n]
                # (int(v) truncates like the reference's int(); the rows were rounded on the device by scale_boxes().round())
                boxes, scores, labels = r[:, :4].astype(np.int64).tolist(), r[:, 4].tolist(), r[:, 5].astype(np.int64).tolist()
                out.append([{"box": b, "score": sc, "label": c, "label_name": names[c], "task": task_of[c]} for b, sc, c in zip(boxes, scores, labels)])
            self._done = out
            self._rows = self._cnt = None
        return self._done


class CerberusDetInference:
    def __init__(self, weights, device: str = "", conf_thres: float = 0.25, iou_thres: float = 0.45, iou_thres_between_tasks: float = 0.8,
                 half: bool = False, img_size: int = 640, full_precision: bool = False):
        """Same arguments as the reference (cerberusdet_inference.py:24-28) plus `full_precision`: the reference's half=False runs the model in
        fp32; here half=False means bf16 storage (the engine's default) unless full_precision=True selects the fp32-accurate path
        (models/cerberus.py::full_precision, cerberusdet_amd/precise.py: boxes within 1e-3 of the fp32 reference, ~8x the time)."""
        self.conf_thres, self.iou_thres, self.iou_thres_between_tasks = conf_thres, iou_thres, iou_thres_between_tasks
        if device and torch.device(device).type != "cuda":
            # the reference would run on the CPU here (cerberusdet_inference.py:30-36, select_device); this engine has no CPU path
            raise RuntimeError(f"CerberusDetInference (cerberusdet_amd): device={device!r} is not supported -- the engine runs on an MI355X only "
                               "(pass '' or 'cuda:N')")
        if not torch.cuda.is_available():
            raise RuntimeError("CerberusDetInference (cerberusdet_amd) needs an MI355X: there is no CPU path")
        self.device = torch.device(device if device else "cuda:0")
        self.half = half
        self.model: CerberusDet = attempt_load(weights, map_location=self.device)
        if full_precision:
            if half:
                raise ValueError("CerberusDetInference: half=True and full_precision=True contradict each other")
            self.model.full_precision()
        elif self.half:
            self.model.half()
        self.model.eval()
        self.stride = int(self.model.stride.max())
        self.names: Dict[str, List[str]] = getattr(self.model, "names", None) or {t: [str(i) for i in range(self.model.get_head(t).nc)]
                                                                                  for t in self.model.heads}
        self.categories_inds_map, self.all_class_names = self._get_categories_map(self.names)
        self._task_of_label = None
        dummy = check_img_size(img_size, s=self.stride)
        self.model(torch.zeros(1, 3, dummy, dummy, device=self.device, dtype=torch.float16 if self.half else torch.float32))  # warm-up

    @staticmethod
    def _get_categories_map(class_names: Dict[str, List[str]]):
        cmap, all_names, last = {}, [], 0
        for task, cats in class_names.items():
            cmap[task] = {i: i + last for i in range(len(cats))}
            last += len(cats)
            all_names.extend(cats)
        return cmap, all_names

    @torch.no_grad()
    def predict(self, tensor: torch.Tensor, original_shape: Union[Tuple[int, int], List[Tuple[int, int]], None] = None, max_det: int = 300,
                agnostic_nms: bool = False, conf_thres: float = None, iou_thres: float = None,
                iou_thres_between_tasks: float = None) -> List[List[Dict]]:
        conf_thres = self.conf_thres if conf_thres is None else conf_thres
        iou_thres = self.iou_thres if iou_thres is None else iou_thres
        iou_bt = self.iou_thres_between_tasks if iou_thres_between_tasks is None else iou_thres_between_tasks
        return self.predict_async(tensor, original_shape, max_det, agnostic_nms, conf_thres, iou_thres, iou_bt).result()

    @torch.no_grad()
    def predict_async(self, tensor: torch.Tensor, original_shape=None, max_det: int = 300, agnostic_nms: bool = False, conf_thres: float = None,
                      iou_thres: float = None, iou_thres_between_tasks: float = None) -> PendingPrediction:
        """`predict` without the wait: everything is enqueued on the current stream (forward, per-task NMS, cross-task merge, the copy
        into pinned host memory); the caller may enqueue the next batch before asking for `.result()`. The forward's plan-owned
        outputs are consumed by the NMS launches of the same call, in stream order, before a later forward can overwrite them."""
        conf_thres = self.conf_thres if conf_thres is None else conf_thres
        iou_thres = self.iou_thres if iou_thres is None else iou_thres
        iou_bt = self.iou_thres_between_tasks if iou_thres_between_tasks is None else iou_thres_between_tasks
        all_out = self.model(tensor.to(self.device), zero_copy=True)  # consumed by NMS right away
        return self.postprocess_async({t: o[0] for t, o in all_out.items()}, tuple(tensor.shape[2:]), original_shape, max_det, agnostic_nms,
                                      conf_thres, iou_thres, iou_bt)

    def predict_stream(self, batches, depth: int = 2, **kwargs):
        """Generator over an iterable of batches -- tensors, or (tensor, original_shape) pairs -- yielding `predict`'s result for each, in
        order, with up to `depth` batches in flight: the result dicts of batch i are built while the GPU runs batch i + 1."""
        from collections import deque

        assert depth >= 1
        pending = deque()
        for b in batches:
            tensor, shape = b if isinstance(b, (tuple, list)) else (b, None)
            pending.append(self.predict_async(tensor, original_shape=shape, **kwargs))
            while len(pending) >= depth + 1:
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()

    def postprocess(self, y_per_task: Dict[str, torch.Tensor], net_shape, original_shape=None, max_det=300, agnostic_nms=False,
                    conf_thres=0.25, iou_thres=0.45, iou_thres_between_tasks=0.8) -> List[List[Dict]]:
        return self.postprocess_async(y_per_task, net_shape, original_shape, max_det, agnostic_nms, conf_thres, iou_thres, iou_thres_between_tasks).result()

    def postprocess_async(self, y_per_task: Dict[str, torch.Tensor], net_shape, original_shape=None, max_det=300, agnostic_nms=False,
                          conf_thres=0.25, iou_thres=0.45, iou_thres_between_tasks=0.8) -> PendingPrediction:
        from . import ops

        tasks = list(y_per_task.keys())
        dev = next(iter(y_per_task.values())).device
        bs = next(iter(y_per_task.values())).shape[0]
        # per-task batched NMS, then class remap + cross-task suppression + scale_boxes().round() in ONE more launch; a single
        # device->host copy per batch feeds the result dicts
        # (the tasks' NMS launch sequences are independent and latency-bound -- one workgroup per image walks its candidates -- so every task
        #  but the first runs on a lane stream beside it: 2 x 0.9 ms -> 0.9 ms per batch of 32 with two heads)
        from .engine import lane_stream

        cur = torch.cuda.current_stream(dev)
        nms = {}
        if len(tasks) > 1 and dev.type == "cuda":
            fork = torch.cuda.Event()
            fork.record(cur)
            for k, t in enumerate(tasks[1:], 1):
                side = lane_stream(dev, k)
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    y = y_per_task[t].contiguous()
                    nms[t] = ops.nms_batched(y, conf_thres, iou_thres, agnostic=agnostic_nms, max_det=max_det)
                y.record_stream(side)  # plan-owned or the caller's: not to be reused before the lane has read it
        nms[tasks[0]] = ops.nms_batched(y_per_task[tasks[0]].contiguous(), conf_thres, iou_thres, agnostic=agnostic_nms, max_det=max_det)
        for k, t in enumerate(tasks[1:], 1):
            if t in nms and len(tasks) > 1 and dev.type == "cuda":
                cur.wait_stream(lane_stream(dev, k))
                for o in nms[t]:
                    o.record_stream(cur)  # allocated on the lane, consumed by the merge on the caller's stream
        rows, cnts = zip(*(nms[t] for t in tasks))
        offs = [self.categories_inds_map[t][0] for t in tasks]  # local id -> global id is a per-task offset (cerberusdet_inference.py:56-70)
        scale = None
        if original_shape is not None:
            shapes = original_shape if isinstance(original_shape, list) else [original_shape] * bs
            sc = []
            for shp in shapes:
                gain = min(net_shape[0] / shp[0], net_shape[1] / shp[1])
                sc.append([gain, (net_shape[1] - shp[1] * gain) / 2, (net_shape[0] - shp[0] * gain) / 2, shp[0], shp[1]])
            # pinned + non_blocking: a pageable host->device copy would wait for everything already enqueued on the stream (the forward)
            scale = torch.tensor(sc, dtype=torch.float32).pin_memory().to(dev, non_blocking=True)
        out, cnt = ops.merge_tasks(rows, cnts, offs, iou_thres_between_tasks, scale)
        h_out = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        h_cnt = torch.empty(cnt.shape, dtype=cnt.dtype, pin_memory=True)
        h_out.copy_(out, non_blocking=True)
        h_cnt.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        st = torch.cuda.current_stream(dev)
        out.record_stream(st)
        cnt.record_stream(st)
        if getattr(self, "_task_of_label", None) is None:  # global label id -> task name (cerberusdet_inference.py:72-83 get_task_by_label)
            task_of = ["unknown"] * len(self.all_class_names)
            for t, ids in self.categories_inds_map.items():
                for g in ids.values():
                    task_of[g] = t
            self._task_of_label = task_of
        return PendingPrediction(h_out, h_cnt, ev, self.all_class_names, self._task_of_label)
