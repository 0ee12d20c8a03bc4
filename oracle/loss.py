"""Oracle: CPU restatement of the detection loss + task-aligned assigner (TEST INFRASTRUCTURE).

Follows (reference paths relative to /root/reference/cerberusdet):
  * Loss.__call__ / preprocess / bbox_decode ...... utils/loss.py:133-181, 111-124, 126-131
  * BboxLoss.forward / _df_loss .................... utils/loss.py:18-44
  * TaskAlignedAssigner + helpers .................. utils/tal.py:13-53, 56-178
  * bbox_iou(xywh=False, CIoU=True) ................ utils/metrics.py:373-412
  * bbox2dist ...................................... utils/tal.py:208-211

Tie rule (documented deviation from an implementation-defined detail): the reference takes
`torch.topk(metrics, 10)` whose order among equal values is implementation-defined (SURVEY.md
section 7 "hard parts"). The oracle -- and the HIP kernel -- break ties by LOWEST ANCHOR INDEX.
`argmax` ties (first maximum wins) already are deterministic in torch and are kept.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .graph import REG_MAX, dist2bbox, make_anchors

TOPK = 10  # utils/loss.py:86
ALPHA = 0.5  # utils/loss.py:88
BETA = 6.0  # utils/loss.py:89
TAL_EPS = 1e-9  # utils/tal.py:58
IOU_EPS = 1e-7  # utils/metrics.py:373


def ciou(box1, box2, eps=IOU_EPS):
    """utils/metrics.py:373-412 with xywh=False, CIoU=True. Broadcasts over leading dims; last dim 4."""
    b1_x1, b1_y1, b1_x2, b1_y2 = box1.chunk(4, -1)
    b2_x1, b2_y1, b2_x2, b2_y2 = box2.chunk(4, -1)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    inter = (torch.minimum(b1_x2, b2_x2) - torch.maximum(b1_x1, b2_x1)).clamp(0) * (
        torch.minimum(b1_y2, b2_y2) - torch.maximum(b1_y1, b2_y1)
    ).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(b1_x2, b2_x2) - torch.minimum(b1_x1, b2_x1)
    ch = torch.maximum(b1_y2, b2_y2) - torch.minimum(b1_y1, b2_y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def pad_targets(batch_idx, cls, prob, bboxes, batch_size, scale_wh):
    """Loss.preprocess (utils/loss.py:111-124): [n,*] labels -> [bs, n_max, 6] (cls, prob, xyxy px)."""
    n = batch_idx.shape[0]
    if n == 0:
        return torch.zeros(batch_size, 0, 6)
    bi = batch_idx.long()
    counts = torch.bincount(bi, minlength=batch_size)
    out = torch.zeros(batch_size, int(counts.max()), 6)
    t = torch.cat((cls.view(-1, 1).float(), prob.view(-1, 1).float(), bboxes.float()), 1)
    for j in range(batch_size):
        m = bi == j
        k = int(m.sum())
        if k:
            out[j, :k] = t[m]
    xywh = out[..., 2:6] * scale_wh
    out[..., 2] = xywh[..., 0] - xywh[..., 2] / 2
    out[..., 3] = xywh[..., 1] - xywh[..., 3] / 2
    out[..., 4] = xywh[..., 0] + xywh[..., 2] / 2
    out[..., 5] = xywh[..., 1] + xywh[..., 3] / 2
    return out


def _topk_lowest_index(metrics, k):
    """top-k along last dim, ties -> lowest index first (stable descending sort)."""
    order = torch.sort(metrics, dim=-1, descending=True, stable=True)[1]
    return order[..., :k]


@torch.no_grad()
def tal_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, num_classes,
               topk=TOPK, alpha=ALPHA, beta=BETA, eps=TAL_EPS):
    """TaskAlignedAssigner.forward (utils/tal.py:67-109).

    pd_scores [b,A,nc] (sigmoid), pd_bboxes [b,A,4] xyxy px, anc_points [A,2] px,
    gt_labels [b,n,1], gt_bboxes [b,n,4], mask_gt [b,n,1].
    Returns target_labels [b,A] i64, target_bboxes [b,A,4], target_scores [b,A,nc],
            fg_mask [b,A] bool, target_gt_idx [b,A] i64.
    """
    bs, na = pd_scores.shape[:2]
    n_max = gt_bboxes.shape[1]
    if n_max == 0:  # utils/tal.py:88-92
        return (torch.full((bs, na), num_classes, dtype=torch.int64), torch.zeros_like(pd_bboxes),
                torch.zeros_like(pd_scores), torch.zeros(bs, na, dtype=torch.bool),
                torch.zeros(bs, na, dtype=torch.int64))
    # get_box_metrics (tal.py:123-133)
    lab = gt_labels.long().squeeze(-1)  # [b,n]
    bbox_scores = torch.gather(pd_scores.permute(0, 2, 1), 1, lab.unsqueeze(-1).expand(-1, -1, na))  # [b,n,A]
    overlaps = ciou(gt_bboxes.unsqueeze(2), pd_bboxes.unsqueeze(1)).squeeze(3).clamp(0)  # [b,n,A]
    align_metric = bbox_scores.pow(alpha) * overlaps.pow(beta)
    # select_candidates_in_gts (tal.py:13-27)
    lt, rb = gt_bboxes.view(-1, 1, 4).chunk(2, 2)
    deltas = torch.cat((anc_points[None] - lt, rb - anc_points[None]), dim=2).view(bs, n_max, na, -1)
    mask_in_gts = (deltas.amin(3) > eps).to(pd_scores.dtype)
    # select_topk_candidates (tal.py:135-154)
    idx = _topk_lowest_index(align_metric * mask_in_gts, topk)  # [b,n,k]
    valid = mask_gt.bool().expand(-1, -1, topk)
    idx = torch.where(valid, idx, torch.zeros_like(idx))
    is_in_topk = torch.zeros(bs, n_max, na, dtype=torch.int64)
    is_in_topk.scatter_add_(2, idx, torch.ones_like(idx))
    is_in_topk = torch.where(is_in_topk > 1, torch.zeros_like(is_in_topk), is_in_topk).to(pd_scores.dtype)
    mask_pos = is_in_topk * mask_in_gts * mask_gt
    # select_highest_overlaps (tal.py:30-53)
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg.unsqueeze(1) > 1).expand(-1, n_max, -1)
        max_idx = overlaps.argmax(1)
        is_max = F.one_hot(max_idx, n_max).permute(0, 2, 1).to(overlaps.dtype)
        mask_pos = torch.where(multi, is_max, mask_pos)
        fg = mask_pos.sum(-2)
    target_gt_idx = mask_pos.argmax(-2)
    # get_targets (tal.py:156-178)
    flat = target_gt_idx + torch.arange(bs).view(-1, 1) * n_max
    target_labels = gt_labels.long().flatten()[flat]
    target_bboxes = gt_bboxes.reshape(-1, 4)[flat]
    target_scores = F.one_hot(target_labels.clamp(0), num_classes)
    target_scores = torch.where(fg[:, :, None] > 0, target_scores, torch.zeros_like(target_scores))
    # normalize (tal.py:102-107)
    align_metric = align_metric * mask_pos
    pos_align = align_metric.amax(-1, keepdim=True)
    pos_over = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (align_metric * pos_over / (pos_align + eps)).amax(-2).unsqueeze(-1)
    target_scores = target_scores * norm
    return target_labels, target_bboxes, target_scores, fg.bool(), target_gt_idx


def detection_loss(feats, batch, nc, gains, strides=(8.0, 16.0, 32.0), return_assign=False):
    """Loss.__call__ (utils/loss.py:133-181) for one task.

    feats: 3 maps [bs, 64+nc, h, w]; batch: dict(batch_idx[n], cls[n,1], prob[n,1], bboxes[n,4] xywh in [0,1]);
    gains: dict(box=, cls=, dfl=). Returns (scalar = 2*bs*sum(box,cls,dfl) [sic, loss.py:179-181], items[4]).
    """
    bs = feats[0].shape[0]
    no = nc + 4 * REG_MAX
    dtype = feats[0].dtype
    x = torch.cat([f.reshape(bs, no, -1) for f in feats], 2)
    pred_distri, pred_scores = x.split((4 * REG_MAX, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()
    pred_distri = pred_distri.permute(0, 2, 1).contiguous()
    h, w = feats[0].shape[2:]
    imgsz = torch.tensor([h, w], dtype=dtype) * strides[0]
    anchor_points, stride_tensor = make_anchors([f.shape[2:] for f in feats], strides, dtype=dtype)

    targets = pad_targets(batch["batch_idx"], batch["cls"], batch["prob"], batch["bboxes"], bs,
                          imgsz[[1, 0, 1, 0]])
    gt_labels, _, gt_bboxes = targets.split((1, 1, 4), 2)
    mask_gt = (gt_bboxes.sum(2, keepdim=True) > 0).to(gt_bboxes.dtype)  # loss.py:155

    b, a, c = pred_distri.shape
    proj = torch.arange(REG_MAX, dtype=dtype)
    dist = pred_distri.view(b, a, 4, c // 4).softmax(3).matmul(proj)  # loss.py:129
    pred_bboxes = dist2bbox(dist, anchor_points, xywh=False)

    tl, tb, ts, fg, tgi = tal_assign(
        pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_tensor).to(gt_bboxes.dtype),
        anchor_points * stride_tensor, gt_labels, gt_bboxes, mask_gt, nc)
    tss = max(ts.sum(), 1)  # loss.py:164

    loss = torch.zeros(4)
    loss_cls = F.binary_cross_entropy_with_logits(pred_scores, ts.to(dtype), reduction="none").sum() / tss
    loss_box = torch.zeros(())
    loss_dfl = torch.zeros(())
    if fg.sum():  # loss.py:171-174
        tbg = tb / stride_tensor
        weight = ts.sum(-1)[fg].unsqueeze(-1)
        iou = ciou(pred_bboxes[fg], tbg[fg])
        loss_box = ((1.0 - iou) * weight).sum() / tss
        # DFL (loss.py:27-44), reg_max-1 = 15 bins upper clamp (tal.py:208-211)
        x1y1, x2y2 = torch.split(tbg, 2, -1)
        t_ltrb = torch.cat((anchor_points - x1y1, x2y2 - anchor_points), -1).clamp(0, REG_MAX - 1 - 0.01)
        pd = pred_distri[fg].view(-1, REG_MAX)
        t = t_ltrb[fg]
        tl_ = t.long()
        tr_ = tl_ + 1
        wl = tr_ - t
        wr = 1 - wl
        dfl = (F.cross_entropy(pd, tl_.view(-1), reduction="none").view(tl_.shape) * wl
               + F.cross_entropy(pd, tr_.view(-1), reduction="none").view(tl_.shape) * wr).mean(-1, keepdim=True)
        loss_dfl = (dfl * weight).sum() / tss
    items = torch.stack((loss_box * gains["box"], loss_cls * gains["cls"], loss_dfl * gains["dfl"]))
    total = items.sum()
    # reference: loss[3] = loss.sum(); return loss.sum() * bs  ->  2 * bs * (box+cls+dfl)
    scalar = (total + total) * bs
    out_items = torch.cat((items, total.view(1))).detach()
    if return_assign:
        return scalar, out_items, dict(target_labels=tl, target_bboxes=tb, target_scores=ts, fg_mask=fg,
                                       target_gt_idx=tgi, pred_bboxes=pred_bboxes.detach())
    return scalar, out_items
