"""Box utilities and NMS with the reference's signatures (ai-forever/CerberusDet cerberusdet/utils/general.py:
206-208 make_divisible, 122-127 check_img_size, 211-213 one_cycle, 272-288 xywh2xyxy, 313-357 scale/clip_boxes,
360-481 non_max_suppression, 484-554 nms_between_tasks). NMS runs as ONE batched gfx950 launch sequence for the whole
batch (csrc/nms.hip) with a single device->host sync for the per-image counts."""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch


def make_divisible(x, divisor):
    """Smallest multiple of `divisor` that is >= x (reference general.py:206-208)."""
    q, r = divmod(x, divisor)
    return int(q + (1 if r else 0)) * divisor


def check_img_size(img_size, s=32):
    """Image side rounded up to the model stride (reference general.py:122-127; the warning print is left out)."""
    return make_divisible(img_size, int(s))


def one_cycle(y1=0.0, y2=1.0, steps=100):
    """Cosine ramp from y1 (x = 0) to y2 (x = steps), the reference's lr schedule factor (general.py:211-213)."""
    span = y2 - y1

    def ramp(x):
        return y1 + span * 0.5 * (1.0 - math.cos(math.pi * x / steps))

    return ramp


def _cat(parts, like):
    return torch.cat(parts, dim=-1) if isinstance(like, torch.Tensor) else np.concatenate(parts, axis=-1)


def xywh2xyxy(x):
    """[..., (cx, cy, w, h, extra...)] -> [..., (x1, y1, x2, y2, extra...)] as a new array (reference general.py:272-288)."""
    centre, half = x[..., 0:2], x[..., 2:4] / 2
    return _cat([centre - half, centre + half, x[..., 4:]], x)


def clip_boxes(boxes, shape):
    """Clamp xyxy boxes in place to an image of shape (h, w) (reference general.py:345-357)."""
    h, w = shape[0], shape[1]
    for col, hi in ((0, w), (1, h), (2, w), (3, h)):
        if isinstance(boxes, torch.Tensor):
            boxes[..., col].clamp_(0, hi)
        else:
            np.clip(boxes[..., col], 0, hi, out=boxes[..., col])


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """Map xyxy boxes (in place) from the letterboxed frame img1_shape back to the original frame img0_shape: remove the padding,
    divide by the resize gain, clamp (reference general.py:313-342)."""
    if ratio_pad is not None:
        gain, (pad_x, pad_y) = ratio_pad[0][0], ratio_pad[1]
    else:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad_x, pad_y = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    for col, pad in ((0, pad_x), (1, pad_y), (2, pad_x), (3, pad_y)):
        boxes[..., col] = (boxes[..., col] - pad) / gain
    clip_boxes(boxes, img0_shape)
    return boxes


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, labels=(),
                        max_det=300, nm=0) -> List[torch.Tensor]:
    """Same contract as the reference: prediction [bs, 4+nc+nm, A] (or the eval tuple (y, feats)) -> list of [k, 6+nm] fp32
    tensors (x1,y1,x2,y2,conf,cls[,mask coefficients]) on the prediction's device. `labels` (autolabelling, general.py:430-436): per
    image a [n, 5] tensor (cls, x, y, w, h) whose rows join the candidates with confidence 1.0 -- here as extra anchors behind the
    model's, which is where the reference's `torch.cat((x, v), 0)` puts them (the order decides ties; the reference's own branch raises as
    written -- it builds the rows nc + nm + 5 columns wide, a YOLOv5 leftover, against candidates of nc + nm + 4 -- so this is its
    intent). `nm` (general.py:410,443-449):
    the last nm channels are mask coefficients that take no part in the suppression and ride along with the kept boxes (zeros for label
    rows). Not reproduced on purpose: the wall-clock `time_limit` bail-out (general.py:417,477-479)."""
    from .. import ops

    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if not prediction.is_cuda:
        raise RuntimeError("cerberusdet_amd.non_max_suppression needs a tensor on the MI355X (no CPU path)")
    masks = None
    if nm:
        mi = prediction.shape[1] - nm  # mask start index = 4 + nc
        assert nm > 0 and mi > 4, f"nm = {nm} leaves no class channel in a prediction of {prediction.shape[1]} channels"
        prediction, masks = prediction[:, :mi], prediction[:, mi:]
    A = prediction.shape[2]
    if labels and any(len(lb) for lb in labels):
        bs, no, _ = prediction.shape
        n_lab = max(len(lb) for lb in labels)
        extra = torch.zeros((bs, no, n_lab), dtype=prediction.dtype, device=prediction.device)  # zero scores: never candidates
        for xi, lb in enumerate(labels):
            if len(lb):
                lb = torch.as_tensor(lb, device=prediction.device).float()
                extra[xi, :4, :len(lb)] = lb[:, 1:5].T.to(prediction.dtype)
                extra[xi, 4 + lb[:, 0].long(), torch.arange(len(lb), device=prediction.device)] = 1.0
        prediction = torch.cat((prediction, extra), 2)
    if masks is None:
        rows, cnt = ops.nms_batched(prediction.contiguous(), conf_thres, iou_thres, classes, agnostic, multi_label, max_det)
        counts = cnt.tolist()  # the only host sync of the call
        return [rows[i, :k] for i, k in enumerate(counts)]
    rows, cnt, anchor = ops.nms_batched(prediction.contiguous(), conf_thres, iou_thres, classes, agnostic, multi_label, max_det, return_anchor=True)
    counts = cnt.tolist()
    out = []
    for i, k in enumerate(counts):
        idx = anchor[i, :k].long()
        m = masks[i].float().T[idx.clamp(max=A - 1)]  # [k, nm]
        m = m * (idx < A).unsqueeze(1)                # label rows carry zero coefficients (general.py:432)
        out.append(torch.cat((rows[i, :k], m), 1))
    return out


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU of xyxy boxes: [n,4] x [m,4] -> [n,m]."""
    lt = torch.maximum(box1[:, None, :2], box2[None, :, :2])
    rb = torch.minimum(box1[:, None, 2:4], box2[None, :, 2:4])
    wh = (rb - lt).clamp_(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    area2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    return inter / (area1[:, None] + area2[None, :] - inter + eps)


def nms_between_tasks(bboxes: torch.Tensor, categories_map_per_task: Dict[str, Dict[int, int]], iou_thres: float) -> torch.Tensor:
    """Cross-task suppression on the rows that survive per-task NMS (reference general.py:484-554): rows are regrouped by task,
    IoU is evaluated between boxes of DIFFERENT tasks only, then a greedy row scan deletes, for each live row with hits, everything
    but the best-scoring box of hits U {row}. Runs on the GPU (cdet_merge_tasks, one workgroup per image); this wrapper keeps the
    reference's single-image signature (rows [n,6] with GLOBAL class ids) -- CerberusDetInference batches the call itself."""
    from .. import ops

    n = bboxes.shape[0]
    if n == 0:
        return bboxes
    dev = bboxes.device if bboxes.is_cuda else torch.device("cuda", torch.cuda.current_device())
    b = bboxes.detach().to(dev, torch.float32)
    cls = b[:, 5].to(torch.int64)
    tasks = list(categories_map_per_task.keys())
    groups = []
    for t in tasks:
        ids = torch.tensor(sorted(categories_map_per_task[t].values()), dtype=torch.int64, device=dev)
        groups.append(torch.nonzero(torch.isin(cls, ids)).flatten())
    max_det = max(max(int(g.numel()) for g in groups), 1)
    rows, cnts = [], []
    for g in groups:
        r = torch.zeros((1, max_det, 6), dtype=torch.float32, device=dev)
        r[0, :g.numel()] = b[g]
        rows.append(r)
        cnts.append(torch.tensor([g.numel()], dtype=torch.int32, device=dev))
    out, cnt = ops.merge_tasks(rows, cnts, [0] * len(tasks), iou_thres, None)  # class ids are already global: offsets 0
    return out[0, :int(cnt[0])].to(bboxes.device)
