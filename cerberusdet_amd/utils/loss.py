"""`Loss` with the reference's interface (ai-forever/CerberusDet cerberusdet/utils/loss.py:48-181):
`Loss(model, task_ids)`; `loss(preds, batch, task) -> (scalar, items[4])` where scalar = 2*bs*(box+cls+dfl) [sic,
loss.py:179-181]. The whole computation -- anchors, DFL decode, task-aligned assignment, BCE, CIoU, DFL and the gradient
w.r.t. the head maps -- is one call into csrc/det_loss.hip; autograd sees a single node."""
from __future__ import annotations

import torch

from .torch_utils import get_hyperparameter


def pad_targets(batch, batch_size, imgsz_hw, device, n_max=None, dropped=None):
    """Loss.preprocess (reference loss.py:111-124) as ONE kernel (csrc/det_loss.hip, cdet_pad_targets):
    labels [n] -> gt [bs, n_max, 5] (cls, x1, y1, x2, y2 in pixels), zero rows = padding, file order kept inside an image.
    n_max=None: sized from the batch like the reference (`counts.max()`, loss.py:117 -- one host sync). With n_max given there is
    no sync; a label that does not fit is dropped and counted in `dropped` (device int32[1]) -- the trainer checks it at its
    once-per-iteration sync, so an undersized n_max fails loudly instead of corrupting targets."""
    import ctypes as C

    from .. import _lib as L
    from ..ops import ptr, stream

    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("cerberusdet_amd.pad_targets runs on the MI355X only (there is no CPU path)")
    bi = batch["batch_idx"].to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    n = bi.numel()
    if n_max is None:
        n_max = int(torch.bincount(bi.long(), minlength=batch_size).max()) if n else 0
    n_max = max(int(n_max), 1)
    cls = batch["cls"].to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    box = batch["bboxes"].to(device=device, dtype=torch.float32).reshape(-1, 4).contiguous()
    out = torch.empty(batch_size, n_max, 5, device=device)
    h, w = imgsz_hw
    lib = L.load()
    L.check(lib.cdet_pad_targets(ptr(bi), ptr(cls), ptr(box), n, batch_size, n_max, float(w), float(h), ptr(out), ptr(dropped), stream()),
            "cdet_pad_targets")
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
