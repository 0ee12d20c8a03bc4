"""Validation metrics with the reference's definitions (utils/metrics.py:48-148, val.py:32-54).

The matching of predictions to labels at the 10 IoU levels runs on the GPU for a whole batch (`cdet_match_predictions`); the
precision/recall curves and AP are accumulated once per evaluation on the host in numpy, as the reference does.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch


def process_batch(detections: torch.Tensor, labels: torch.Tensor, iouv: torch.Tensor) -> torch.Tensor:
    """Single-image signature of the reference (val.py:32): detections [N,6] (x1,y1,x2,y2,conf,cls), labels [M,5]
    (cls,x1,y1,x2,y2), iouv [T] -> correct [N,T] bool on detections.device. `val.run` batches the call itself."""
    from .. import ops

    n, m = detections.shape[0], labels.shape[0]
    out_dev = detections.device
    if n == 0:
        return torch.zeros((0, iouv.numel()), dtype=torch.bool, device=out_dev)
    dev = detections.device if detections.is_cuda else torch.device("cuda", torch.cuda.current_device())
    rows = detections.detach().to(dev, torch.float32).reshape(1, n, 6).contiguous()
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    start = torch.tensor([0, m], dtype=torch.int32, device=dev)
    correct = ops.match_predictions(rows, cnt, labels.detach().to(dev, torch.float32), start, iouv.to(dev), max_labels=max(m, 1))
    return correct[0].bool().to(out_dev)


def smooth(y: np.ndarray, f: float = 0.05) -> np.ndarray:
    """Box filter over a fraction f of the curve, edge-padded."""
    nf = round(len(y) * f * 2) // 2 + 1
    pad = np.ones(nf // 2)
    return np.convolve(np.concatenate((pad * y[0], y, pad * y[-1])), np.ones(nf) / nf, mode="valid")


def compute_ap(recall: np.ndarray, precision: np.ndarray) -> Tuple[float, np.ndarray, np.ndarray]:
    """101-point interpolated AP of the precision envelope (COCO style)."""
    mrec = np.concatenate(([0.0], recall, [recall[-1] + 0.01]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    return float(np.sum((y[1:] + y[:-1]) * np.diff(x)) / 2.0), mpre, mrec


def ap_per_class(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, target_cls: np.ndarray, eps: float = 1e-16):
    """tp [n,T] bool/0-1, conf [n], pred_cls [n], target_cls [m] -> (tp, fp, p, r, f1, ap [nc,T], classes) at the confidence
    that maximises the smoothed mean F1 (utils/metrics.py:56-125; plotting omitted)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, nt = np.unique(target_cls, return_counts=True)
    px = np.linspace(0, 1, 1000)
    ap = np.zeros((len(classes), tp.shape[1]))
    p, r = np.zeros((len(classes), 1000)), np.zeros((len(classes), 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        if sel.sum() == 0 or nt[ci] == 0:
            continue
        tpc = tp[sel].cumsum(0)
        fpc = (1 - tp[sel]).cumsum(0)
        recall = tpc / (nt[ci] + eps)
        precision = tpc / (tpc + fpc)
        r[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        p[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1 = 2 * p * r / (p + r + eps)
    k = smooth(f1.mean(0), 0.1).argmax()
    p, r, f1 = p[:, k], r[:, k], f1[:, k]
    tpn = (r * nt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, classes.astype(int)
