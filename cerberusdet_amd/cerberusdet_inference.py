"""`CerberusDetInference` with the reference's constructor / attributes / `predict` contract
(ai-forever/CerberusDet cerberusdet/cerberusdet_inference.py:19-54, 85-186).

Execution differs: the all-heads forward is one compiled gfx950 launch list (backbone + shared neck once, each branch once),
per-task NMS is ONE batched launch sequence per task, and the per-image combine / cross-task NMS / rescale runs on the few
hundred surviving rows after a single device->host copy (the reference does `.cpu()` + python lambdas per task per image).
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


class CerberusDetInference:
    def __init__(self, weights, device: str = "", conf_thres: float = 0.25, iou_thres: float = 0.45, iou_thres_between_tasks: float = 0.8,
                 half: bool = False, img_size: int = 640):
        self.conf_thres, self.iou_thres, self.iou_thres_between_tasks = conf_thres, iou_thres, iou_thres_between_tasks
        if not torch.cuda.is_available():
            raise RuntimeError("CerberusDetInference (cerberusdet_amd) needs an MI355X: there is no CPU path")
        self.device = torch.device(device if device and device != "cpu" else "cuda:0")
        self.half = half
        self.model: CerberusDet = attempt_load(weights, map_location=self.device)
        if self.half:
            self.model.half()
        self.model.eval()
        self.stride = int(self.model.stride.max())
        self.names: Dict[str, List[str]] = getattr(self.model, "names", None) or {t: [str(i) for i in range(self.model.get_head(t).nc)]
                                                                                  for t in self.model.heads}
        self.categories_inds_map, self.all_class_names = self._get_categories_map(self.names)
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
        all_out = self.model(tensor.to(self.device), zero_copy=True)  # consumed by NMS right away
        return self.postprocess({t: o[0] for t, o in all_out.items()}, tuple(tensor.shape[2:]), original_shape, max_det, agnostic_nms,
                                conf_thres, iou_thres, iou_bt)

    def postprocess(self, y_per_task: Dict[str, torch.Tensor], net_shape, original_shape=None, max_det=300, agnostic_nms=False,
                    conf_thres=0.25, iou_thres=0.45, iou_thres_between_tasks=0.8) -> List[List[Dict]]:
        from . import ops

        tasks = list(y_per_task.keys())
        dev = next(iter(y_per_task.values())).device
        bs = next(iter(y_per_task.values())).shape[0]
        # per-task batched NMS, then class remap + cross-task suppression + scale_boxes().round() in ONE more launch; a single
        # device->host copy per batch feeds the result dicts
        rows, cnts = zip(*(ops.nms_batched(y_per_task[t].contiguous(), conf_thres, iou_thres, agnostic=agnostic_nms, max_det=max_det)
                           for t in tasks))
        offs = [self.categories_inds_map[t][0] for t in tasks]  # local id -> global id is a per-task offset (cerberusdet_inference.py:56-70)
        scale = None
        if original_shape is not None:
            shapes = original_shape if isinstance(original_shape, list) else [original_shape] * bs
            sc = []
            for shp in shapes:
                gain = min(net_shape[0] / shp[0], net_shape[1] / shp[1])
                sc.append([gain, (net_shape[1] - shp[1] * gain) / 2, (net_shape[0] - shp[0] * gain) / 2, shp[0], shp[1]])
            scale = torch.tensor(sc, dtype=torch.float32, device=dev)
        out, cnt = ops.merge_tasks(rows, cnts, offs, iou_thres_between_tasks, scale)
        out, cnt = out.cpu(), cnt.cpu().tolist()
        bounds = []
        for t in tasks:
            ids = self.categories_inds_map[t]
            bounds.append((min(ids.values()), max(ids.values()), t))
        results = []
        for i in range(bs):
            img = []
            for row in out[i, :cnt[i]].tolist():
                c = int(row[5])
                task = next((tn for lo, hi, tn in bounds if lo <= c <= hi), "unknown")
                img.append({"box": [int(v) for v in row[:4]], "score": float(row[4]), "label": c, "label_name": self.all_class_names[c],
                            "task": task})
            results.append(img)
        return results
