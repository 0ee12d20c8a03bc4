"""torch.autograd bridge: lets reference-style code (`out = model(img, task); loss(out).backward()`) drive the compiled
plan. Forward returns the three raw head maps (views of the plan's fp32 buffers, NCHW-shaped); backward copies the
incoming gradients into the plan's head-gradient buffers and replays the backward launch list. Parameter gradients are
accumulated IN PLACE into the model-owned fp32 `.grad` buffers (the reference accumulates over the per-task passes of an
iteration too, trainers/averaging.py:142-168), so the Function returns None for them."""
from __future__ import annotations

import torch


class _PlanFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, x, anchor, fresh=False):
        plan.run_forward(x)
        ctx.plan = plan
        ctx.generation = plan.generation  # the saved z / mean / invstd live in the plan's buffers: valid until its next forward
        outs = []
        for t in plan.tasks:
            nc = plan.model.get_head(t).nc
            # fresh: one FLAT copy of each padded NHWC map (memcpy speed), then the NCHW-shaped view of the copy
            outs += [(f.clone() if fresh else f)[..., :64 + nc].permute(0, 3, 1, 2) for f in plan.feats[t]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("backward() of a forward whose plan has run forward again since: the compiled plan keeps ONE set of saved "
                               "activations per (tasks, shape, dtype) -- call backward() before the next forward of the same configuration "
                               "(the reference's per-task forward / backward order does)")
        i = 0
        for t in plan.tasks:
            nc = plan.model.get_head(t).nc
            for d in plan.dfeats[t]:
                g = grads[i]
                i += 1
                if g is None:
                    d.zero_()
                else:
                    d[..., :64 + nc].copy_(g.permute(0, 2, 3, 1))
        plan.run_backward()
        plan.model._merge_alt_grads()  # (a trainer-managed model keeps per-task gradient buckets on its shared blocks: p.grad must hold the sum)
        return None, None, None, None


def run_with_autograd(plan, x, fresh=False):
    # `anchor` is a leaf that requires grad so that autograd records the node even though the image does not need gradients
    anchor = plan.model._autograd_anchor()
    outs = _PlanFunction.apply(plan, x, anchor, fresh)
    res, i = {}, 0
    for t in plan.tasks:
        res[t] = list(outs[i:i + 3])
        i += 3
    return res
