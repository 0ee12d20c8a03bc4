"""Oracle: numpy restatement of NMS and cross-task NMS (TEST INFRASTRUCTURE).

Follows (reference paths relative to /root/reference/cerberusdet):
  * non_max_suppression ........ utils/general.py:360-481
  * xywh2xyxy .................. utils/general.py:272-288
  * nms_between_tasks .......... utils/general.py:484-554
  * box_iou .................... utils/metrics.py:415-433
  * scale_boxes / clip_boxes ... utils/general.py:313-357
  * _combine_output / predict .. cerberusdet_inference.py:72-83, 117-184

Third-party boundary -- **parity unpinned**: `torchvision.ops.nms` (torchvision==0.20.1, pinned at
/root/reference/pyproject.toml:52, source not vendored, package not installed here). `greedy_nms`
restates its published algorithm: visit boxes by descending score; a box is kept iff its IoU with
every previously KEPT box is <= thr (suppression is strict `>`); IoU = inter / (area_a + area_b -
inter), area = (x2-x1)*(y2-y1) with no +1 and no eps; returns kept indices by descending score.
Tie rule: equal scores are visited in original index order (stable sort) -- what
`argsort(descending=True)` produced on CPU in the survey probe.

Deliberately NOT reproduced: the wall-clock `time_limit` bail-out (general.py:417,477-479), which
silently truncates the batch and is nondeterministic.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

MAX_WH = 7680.0  # general.py:415
MAX_NMS = 30000  # general.py:416


def greedy_nms(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """torchvision.ops.nms semantics in fp32. boxes [n,4] xyxy, scores [n] -> kept indices (int64)."""
    boxes = boxes.astype(np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    suppressed = np.zeros(n, bool)
    keep = []
    thr = np.float32(iou_thres)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype(np.float32)
        iou = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[iou > thr]] = True
    return np.asarray(keep, np.int64)


def xywh2xyxy(x: np.ndarray) -> np.ndarray:
    y = x.copy()
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def non_max_suppression(prediction: np.ndarray, conf_thres=0.25, iou_thres=0.45, classes: Optional[Sequence[int]] = None,
                        agnostic=False, multi_label=False, max_det=300, nm=0) -> List[np.ndarray]:
    """utils/general.py:360-481 (labels=()). prediction [bs, 4+nc+nm, A] (any float dtype; the
    reference promotes to fp32 through `j.float()` in the cat, general.py:446-449, so everything after
    the candidate filter is fp32). Returns list of [k,6+nm] fp32 rows (x1,y1,x2,y2,conf,cls[,mask coefficients]):
    the nm channels behind the class scores (general.py:410-411 `mi = 4 + nc`, :443 `x.split((4, nc, nm), 1)`) take no part
    in the suppression and are concatenated to the kept rows (:447, :450)."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    bs, no, _ = prediction.shape
    nc = no - nm - 4
    mi = 4 + nc
    multi_label = multi_label and nc > 1
    thr = prediction.dtype.type(conf_thres)
    out = []
    for xi in range(bs):
        x = prediction[xi].T  # [A, 4+nc+nm]
        xc = x[:, 4:mi].max(1) > thr
        x = x[xc]
        if not x.shape[0]:
            out.append(np.zeros((0, 6 + nm), np.float32))
            continue
        box = xywh2xyxy(x[:, :4])
        cls = x[:, 4:mi]
        mask = x[:, mi:].astype(np.float32)
        if multi_label:
            i, j = np.nonzero(cls > thr)
            rows = np.concatenate((box[i].astype(np.float32), cls[i, j, None].astype(np.float32),
                                   j[:, None].astype(np.float32), mask[i]), 1)
        else:
            j = cls.argmax(1)  # first max on ties, like torch.max(1)
            conf = cls[np.arange(cls.shape[0]), j]
            rows = np.concatenate((box.astype(np.float32), conf[:, None].astype(np.float32),
                                   j[:, None].astype(np.float32), mask), 1)
            rows = rows[conf > thr]
        if classes is not None:
            rows = rows[np.isin(rows[:, 5], np.asarray(classes, np.float32))]
        if not rows.shape[0]:
            out.append(np.zeros((0, 6 + nm), np.float32))
            continue
        order = np.argsort(-rows[:, 4], kind="stable")[:MAX_NMS]
        rows = rows[order]
        c = rows[:, 5:6] * np.float32(0 if agnostic else MAX_WH)
        keep = greedy_nms(rows[:, :4] + c, rows[:, 4], iou_thres)[:max_det]
        out.append(rows[keep])
    return out


def box_iou(box1: np.ndarray, box2: np.ndarray, eps=1e-7) -> np.ndarray:  # utils/metrics.py:415-433
    a1, a2 = box1[:, None, :2], box1[:, None, 2:4]
    b1, b2 = box2[None, :, :2], box2[None, :, 2:4]
    inter = np.clip(np.minimum(a2, b2) - np.maximum(a1, b1), 0, None).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + np.float32(eps))


def nms_between_tasks(bboxes: np.ndarray, categories_map_per_task: Dict[str, Dict[int, int]], iou_thres: float):
    """utils/general.py:484-554. bboxes [n,6] with GLOBAL class ids. Returns surviving rows
    (re-ordered grouped by task, as the reference does)."""
    tasks = list(categories_map_per_task.keys())
    order, sizes = [], []
    for t in tasks:
        ids = set(categories_map_per_task[t].values())
        inds = [i for i in range(bboxes.shape[0]) if int(bboxes[i, 5]) in ids]
        order += inds
        sizes.append(len(inds))
    n = bboxes.shape[0]
    iou = np.zeros((n, n), np.float32)  # general.py:493 allocates with the ORIGINAL count
    b = bboxes[order]
    starts = np.concatenate(([0], np.cumsum(sizes)))
    for i in range(len(tasks)):
        if sizes[i] == 0:
            continue
        for j in range(i + 1, len(tasks)):
            if sizes[j] == 0:
                continue
            bi = b[starts[i]:starts[i + 1], :4]
            bj = b[starts[j]:starts[j + 1], :4]
            iou[starts[i]:starts[i + 1], starts[j]:starts[j + 1]] = box_iou(bi, bj)
    if not (iou > iou_thres).any():
        return b
    to_delete = set()
    for r in range(iou.shape[0]):
        if r in to_delete:
            continue
        idxs = np.nonzero(iou[r] > iou_thres)[0]
        if len(idxs) == 0:
            continue
        idxs = np.concatenate((idxs, [r]))
        best = int(np.argmax(b[idxs, 4]))  # first max on ties
        to_delete.update(int(idxs[k]) for k in range(len(idxs)) if k != best)
    if len(b) == len(to_delete):
        return b
    keep = [i for i in range(len(b)) if i not in to_delete]
    return b[keep]


def clip_boxes(boxes, shape):  # general.py:343-357
    boxes[..., 0] = boxes[..., 0].clip(0, shape[1])
    boxes[..., 1] = boxes[..., 1].clip(0, shape[0])
    boxes[..., 2] = boxes[..., 2].clip(0, shape[1])
    boxes[..., 3] = boxes[..., 3].clip(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape):  # general.py:313-340 (ratio_pad=None)
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    boxes = boxes.copy()
    boxes[..., [0, 2]] -= np.float32(pad[0])
    boxes[..., [1, 3]] -= np.float32(pad[1])
    boxes[..., :4] /= np.float32(gain)
    return clip_boxes(boxes, img0_shape)


def categories_map(names: Dict[str, List[str]]):
    """cerberusdet_inference.py:56-70: per-task local class id -> global id (task order offsets)."""
    out, all_names, last = {}, [], 0
    for task, cats in names.items():
        out[task] = {i: i + last for i in range(len(cats))}
        last += len(cats)
        all_names.extend(cats)
    return out, all_names


def predict_postprocess(y_per_task: Dict[str, np.ndarray], names: Dict[str, List[str]], net_shape, original_shape=None,
                        conf_thres=0.25, iou_thres=0.45, iou_thres_between_tasks=0.8, max_det=300, agnostic=False):
    """Everything in CerberusDetInference.predict after the forward (cerberusdet_inference.py:121-184)."""
    cmap, all_names = categories_map(names)
    per_task = {t: non_max_suppression(y, conf_thres, iou_thres, agnostic=agnostic, max_det=max_det)
                for t, y in y_per_task.items()}
    bs = next(iter(y_per_task.values())).shape[0]
    results = []
    for i in range(bs):
        det = np.zeros((0, 6), np.float32)
        for t, lst in per_task.items():
            d = lst[i].copy()
            if d.shape[0]:
                d[:, 5] = np.asarray([cmap[t][int(c)] for c in d[:, 5]], np.float32)
                det = np.concatenate((det, d), 0)
        det = nms_between_tasks(det, cmap, iou_thres_between_tasks)
        if len(det) and original_shape is not None:
            shp = original_shape[i] if isinstance(original_shape, list) else original_shape
            det = det.copy()
            det[:, :4] = np.round(scale_boxes(net_shape, det[:, :4], shp))
        img = []
        for row in det:
            c = int(row[5])
            task = next((t for t, m in cmap.items() if c in m.values()), "unknown")
            img.append(dict(box=[int(v) for v in row[:4]], score=float(row[4]), label=c,
                            label_name=all_names[c], task=task))
        results.append(img)
    return results
