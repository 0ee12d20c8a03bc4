"""Validation pass with the reference's settings and metric definitions (val.py:194-400): eval forward of one task -> batched
NMS at conf 0.001 / IoU 0.6 / multi_label -> boxes and labels to native image space -> device matcher at IoU 0.5:0.95 -> AP.

Scope: the arithmetic between the model output and (mp, mr, mAP50, mAP); data loading, plots, COCO-json export and the per-task
logging of the reference's harness are not reproduced. `batches` are the reference's batch dicts (data/datasets.py:440-459)."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import numpy as np
import torch

from . import ops
from .utils.metrics import ap_per_class


def _to_native(boxes: torch.Tensor, net_hw, shape, ratio_pad=None) -> torch.Tensor:
    """scale_boxes + clip_boxes (utils/general.py:313-357) on [k,4] xyxy, fp32, on the device."""
    if ratio_pad is None:
        gain = min(net_hw[0] / shape[0], net_hw[1] / shape[1])
        pad = ((net_hw[1] - shape[1] * gain) / 2, (net_hw[0] - shape[0] * gain) / 2)
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    b = boxes.clone()
    b[:, [0, 2]] -= pad[0]
    b[:, [1, 3]] -= pad[1]
    b /= gain
    b[:, [0, 2]] = b[:, [0, 2]].clamp(0, shape[1])
    b[:, [1, 3]] = b[:, [1, 3]].clamp(0, shape[0])
    return b


@torch.no_grad()
def run(model, task: str, batches: Iterable[dict], conf_thres: float = 0.001, iou_thres: float = 0.6, max_det: int = 300,
        half: bool = True, single_cls: bool = False, names: Optional[Dict[int, str]] = None) -> dict:
    """One task's validation: returns {"mp", "mr", "map50", "map", "ap" [nc,10], "classes", "seen", "nt"}."""
    device = next(model.parameters()).device
    was_training = model.training
    model.eval()
    iouv = torch.linspace(0.5, 0.95, 10, device=device)
    stats, seen = [], 0
    for batch in batches:
        img = batch["img"].to(device)
        if img.dtype != torch.uint8:
            img = img.half() if half else img.float()
        N, _, H, W = img.shape
        y = model(img, task, zero_copy=True)  # consumed by NMS right away
        y = y[0] if isinstance(y, (tuple, list)) else y
        rows, cnt = ops.nms_batched(y.contiguous(), conf_thres, iou_thres, agnostic=single_cls, multi_label=True, max_det=max_det)
        bi = batch["batch_idx"].to(device).long()
        order = torch.argsort(bi, stable=True)
        bi = bi[order]
        cls = batch["cls"].to(device).float().reshape(-1)[order]
        box = batch["bboxes"].to(device).float()[order]
        xyxy = torch.stack((box[:, 0] - box[:, 2] / 2, box[:, 1] - box[:, 3] / 2, box[:, 0] + box[:, 2] / 2, box[:, 1] + box[:, 3] / 2), 1)
        xyxy = xyxy * torch.tensor((W, H, W, H), device=device, dtype=torch.float32)
        counts = torch.bincount(bi, minlength=N)
        start = torch.zeros(N + 1, dtype=torch.int32, device=device)
        start[1:] = torch.cumsum(counts, 0)
        shapes = batch.get("ori_shape") or [(H, W)] * N
        rps = batch.get("ratio_pad") or [None] * N
        if single_cls:
            rows[..., 5] = 0
        predn = rows.clone()
        labn = xyxy.clone()
        st = start.tolist()
        for si in range(N):
            if tuple(shapes[si]) != (H, W) or rps[si] is not None:
                predn[si, :, :4] = _to_native(rows[si, :, :4], (H, W), shapes[si], rps[si])
                labn[st[si]:st[si + 1]] = _to_native(xyxy[st[si]:st[si + 1]], (H, W), shapes[si], rps[si])
        labels = torch.cat((cls.unsqueeze(1), labn), 1)
        correct = ops.match_predictions(predn.contiguous(), cnt, labels, start, iouv, max_labels=max(int(counts.max()) if N else 0, 1))
        cnt_h = cnt.tolist()
        correct, rows_h, cls_h = correct.cpu().numpy(), rows.cpu().numpy(), cls.cpu().numpy()
        for si in range(N):
            seen += 1
            k, tcls = cnt_h[si], cls_h[st[si]:st[si + 1]]
            if k == 0 and len(tcls) == 0:
                continue
            stats.append((correct[si, :k].astype(bool), rows_h[si, :k, 4], rows_h[si, :k, 5], tcls))
    model.train(was_training)
    out = dict(mp=0.0, mr=0.0, map50=0.0, map=0.0, ap=np.zeros((0, 10)), classes=np.zeros(0, int), seen=seen, nt=np.zeros(0, int))
    if stats:
        tp, conf, pcls, tcls = [np.concatenate(x, 0) for x in zip(*stats)]
        if len(tp) and tp.any():
            _, _, p, r, _, ap, classes = ap_per_class(tp, conf, pcls, tcls)
            out.update(mp=float(p.mean()), mr=float(r.mean()), map50=float(ap[:, 0].mean()), map=float(ap.mean()), ap=ap, classes=classes)
        out["nt"] = np.bincount(tcls.astype(int)) if len(tcls) else np.zeros(0, int)
    return out
