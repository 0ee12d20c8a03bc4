"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's validation arithmetic (numpy).

* process_batch : val.py:32-54        (which predictions count as correct at the 10 IoU levels)
* box_iou       : utils/metrics.py:415-433
* smooth, ap_per_class, compute_ap : utils/metrics.py:48-148

Pinned by tests/golden/val.* (tools/make_golden_val.py runs the real reference). Only tests/ may import this module.

Tie rule (the reference sorts candidate matches with numpy's unstable argsort()[::-1]): when two labels have exactly the same
IoU with a prediction, the label with the HIGHER index is preferred here -- what the reversed stable order gives; the golden
cases are random floats and contain no such tie ("parity unpinned" for exact ties).
"""
import numpy as np


def box_iou(box1: np.ndarray, box2: np.ndarray, eps=1e-7) -> np.ndarray:
    a1, a2 = box1[:, None, :2], box1[:, None, 2:4]
    b1, b2 = box2[None, :, :2], box2[None, :, 2:4]
    inter = np.clip(np.minimum(a2, b2) - np.maximum(a1, b1), 0, None).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + np.float32(eps))


def process_batch(detections: np.ndarray, labels: np.ndarray, iouv: np.ndarray) -> np.ndarray:
    """detections [N,6] (x1,y1,x2,y2,conf,cls), labels [M,5] (cls,x1,y1,x2,y2), iouv [T] -> correct [N,T] bool.
    Per threshold: every prediction proposes its best class-matching label (highest IoU >= thr); a label accepts the proposing
    prediction with the LOWEST index (the reference de-duplicates labels on the prediction-sorted match list, val.py:47-52)."""
    n, m = detections.shape[0], labels.shape[0]
    correct = np.zeros((n, iouv.shape[0]), bool)
    if n == 0 or m == 0:
        return correct
    iou = box_iou(labels[:, 1:].astype(np.float32), detections[:, :4].astype(np.float32))  # [M, N]
    same = labels[:, 0:1] == detections[None, :, 5]
    best_l = np.full(n, -1)
    best_iou = np.full(n, -1.0, np.float32)
    for d in range(n):
        for l in range(m):
            if same[l, d] and iou[l, d] >= best_iou[d]:  # >= : the later (higher) label wins an exact tie
                best_iou[d], best_l[d] = iou[l, d], l
    for t, thr in enumerate(iouv):
        taken = set()
        for d in range(n):  # ascending prediction index
            if best_l[d] >= 0 and best_iou[d] >= thr and best_l[d] not in taken:
                taken.add(best_l[d])
                correct[d, t] = True
    return correct


def smooth(y, f=0.05):
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [recall[-1] + 0.01]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    ap = float(np.sum((y[1:] + y[:-1]) * np.diff(x)) / 2.0)  # np.trapz
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        r[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + eps)
    k = smooth(f1.mean(0), 0.1).argmax()
    p, r, f1 = p[:, k], r[:, k], f1[:, k]
    tpn = (r * nt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, unique_classes.astype(int)
