"""`Loss` with the reference's interface (ai-forever/CerberusDet cerberusdet/utils/loss.py:48-181):
`Loss(model, task_ids)`; `loss(preds, batch, task) -> (scalar, items[4])` where scalar = 2*bs*(box+cls+dfl) [sic,
loss.py:179-181]. The whole computation -- anchors, DFL decode, task-aligned assignment, BCE, CIoU, DFL and the gradient
w.r.t. the head maps -- is one call into csrc/det_loss.hip; autograd sees a single node."""
from __future__ import annotations

import torch

from .torch_utils import get_hyperparameter


def pad_targets(batch, batch_size, imgsz_hw, device, n_max=None):
    """Loss.preprocess (reference loss.py:111-124) without the python loop over images:
    labels [n] -> [bs, n_max, 5] (cls, x1, y1, x2, y2 in pixels), zero rows = padding."""
    bi = batch["batch_idx"].to(device).view(-1).long()
    n = bi.numel()
    if n == 0:
        return torch.zeros(batch_size, 1, 5, device=device)
    cls = batch["cls"].to(device).view(-1).float()
    box = batch["bboxes"].to(device).float()
    order = torch.argsort(bi, stable=True)
    bi, cls, box = bi[order], cls[order], box[order]
    counts = torch.zeros(batch_size, dtype=torch.int64, device=device).index_add_(0, bi, torch.ones_like(bi))
    sync_free = n_max is not None  # caller guarantees n_max >= labels per image: no device->host sync at all
    if n_max is None:
        n_max = int(counts.max())  # one host sync (the reference has `counts.max()` as well, loss.py:117)
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(n, device=device) - start[bi]
    h, w = imgsz_hw
    scale = torch.tensor([w, h, w, h], dtype=torch.float32, device=device)
    xywh = box * scale
    xyxy = torch.stack((xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2), 1)
    out = torch.zeros(batch_size, max(n_max, 1), 5, device=device)
    vals = torch.cat((cls[:, None], xyxy), 1)
    if sync_free:
        out.index_put_((bi, pos.clamp(max=max(n_max, 1) - 1)), vals)
    else:
        out[bi, pos] = vals
    return out


class _DetLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, nc, gains, strides, f0, f1, f2):
        from .. import ops

        feats = [f.permute(0, 2, 3, 1).contiguous().float() for f in (f0, f1, f2)]  # NHWC fp32 (no-op for plan buffers' layout)
        loss5, dfe, _ = ops.det_loss(feats, gt, nc, gains, strides, grad_scale=1.0, grad_dtype=torch.float32)
        ctx.dfe = dfe
        return loss5[4].clone(), loss5[:4].clone()

    @staticmethod
    def backward(ctx, g_scalar, g_items):
        grads = [d.permute(0, 3, 1, 2) * g_scalar for d in ctx.dfe]
        return None, None, None, None, grads[0], grads[1], grads[2]


class Loss:
    def __init__(self, model, task_ids):
        h = model.hyp
        self.device = next(model.parameters()).device
        self.stride, self.reg_max = None, None
        self.nc, self.no, self.loss_weights = {}, {}, {}
        for task_idx, task in enumerate(task_ids):
            self.loss_weights[task] = dict(box=get_hyperparameter(h, "box", task_idx, task), dfl=get_hyperparameter(h, "dfl", task_idx, task),
                                           cls=get_hyperparameter(h, "cls", task_idx, task))
        for task_name in model.heads:
            head = model.get_head(task_name)
            self.nc[task_name], self.no[task_name] = head.nc, head.no
            assert self.reg_max in (None, head.reg_max)
            self.reg_max = head.reg_max
            assert self.stride is None or torch.equal(self.stride, head.stride)
            self.stride = head.stride
        self.use_dfl = self.reg_max > 1

    def __call__(self, preds, batch, task):
        feats = preds[1] if isinstance(preds, tuple) else preds
        bs = feats[0].shape[0]
        imgsz = (feats[0].shape[2] * float(self.stride[0]), feats[0].shape[3] * float(self.stride[0]))
        gt = pad_targets(batch, bs, imgsz, feats[0].device)
        scalar, items = _DetLossFn.apply(gt, self.nc[task], self.loss_weights[task], [float(s) for s in self.stride], *feats)
        return scalar, items.detach()
