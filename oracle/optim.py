"""Oracle: CPU restatement of the multi-task optimizer step semantics (TEST INFRASTRUCTURE).

Follows (reference paths relative to /root/reference/cerberusdet):
  * Averaging.optimizer_step .............. trainers/averaging.py:205-223
  * get_optimizer (param groups) .......... trainers/averaging.py:242-269
  * torch.optim.SGD(nesterov=True) ........ torch semantics (momentum buffer initialised with the first d_p)
  * clip_grad_norm_(max_norm=10) .......... torch semantics: coef = min(1, max/(norm+1e-6))
  * ModelEMA.update ....................... utils/torch_utils.py:302-312
  * warmup_lr ............................. trainers/base_trainer.py:100-112

All functions operate on flat ``{state_dict_key: tensor}`` dicts (reference key schema).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch


def param_group(key: str) -> int:
    """0 = weights with decay, 1 = BN weights (no decay), 2 = biases (no decay). averaging.py:249-262."""
    if key.endswith(".bias"):
        return 2
    if key.endswith("bn.weight"):
        return 1
    return 0


def is_trainable(key: str) -> bool:
    if key.endswith(("running_mean", "running_var", "num_batches_tracked")):
        return False
    return not key.endswith("dfl.conv.weight")  # frozen, models/yolo.py:52


def block_of(key: str) -> int:
    return int(key.split(".")[1])


class GradScalerState:
    """torch.cuda.amp.GradScaler as the reference drives it (trainers/averaging.py:61 `amp.GradScaler(enabled=self.cuda)`; 158 scale(loss).backward();
    207 unscale_; 219 step; 220 update), restated: defaults init_scale 65536, growth_factor 2, backoff_factor 0.5, growth_interval 2000.
    growth_interval <= 0 keeps the scale fixed (the engine's bf16 plans: scale 1, only the found-inf skip applies)."""

    def __init__(self, scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(scale), growth_factor, backoff_factor, growth_interval
        self.growth_tracker, self.skipped = 0, 0

    def update(self, found_inf: bool):  # torch/amp/grad_scaler.py::update -> _amp_update_scale_
        if found_inf:
            self.skipped += 1
            if self.growth_interval > 0:
                self.scale *= self.backoff_factor
                self.growth_tracker = 0
        elif self.growth_interval > 0:
            self.growth_tracker += 1
            if self.growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self.growth_tracker = 0


def optimizer_step(weights: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], momentum_buf: Dict[str, torch.Tensor],
                   n_serving: Dict[int, int], lr=(0.00309, 0.00309, 0.00309), momentum=0.952, weight_decay=0.00037,
                   max_norm=10.0, scaler: "GradScalerState | None" = None):
    """One Averaging.optimizer_step on CPU tensors.

    scaler: None = scale 1 and no skip bookkeeping (the pre-round-6 behaviour for finite gradients). With a GradScalerState the gradients are the
    SCALED ones (scaler.scale(loss).backward()): unscale_ divides them by the scale and records found_inf = any non-finite element; on found_inf
    scaler.step() skips optimizer.step() -- weights and momentum buffers untouched -- and update() backs the scale off (averaging.py:207-220). The
    caller's zero_grad() / ema.update() happen either way. Returns the total norm of the unscaled gradients (non-finite on a skipped step).

    grads: accumulated over this iteration's task passes (keys missing = no grad).
    n_serving: block idx -> number of tasks served (averaging.py:124-127; max(len, 1)).
    lr: per param-group learning rate (group order 0,1,2 as `param_group`).
    Returns the total gradient norm before clipping.
    """
    keys = [k for k in weights if is_trainable(k) and k in grads]
    inv = 1.0 / scaler.scale if scaler is not None else 1.0
    found_inf = any(not bool(torch.isfinite(grads[k]).all()) for k in keys)
    total = torch.sqrt(sum(((grads[k].double() * inv) ** 2).sum() for k in keys)).float()
    if scaler is not None or found_inf:
        if scaler is not None:
            scaler.update(found_inf)
        if found_inf:
            return float(total)  # scaler.step(): optimizer.step() skipped
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for k in keys:
        g = grads[k] * inv * coef  # unscale_, then clip over ALL params (averaging.py:207-208)
        g = g / float(n_serving[block_of(k)])  # then the per-block division (averaging.py:211-217)
        grp = param_group(k)
        p = weights[k]
        if grp == 0 and weight_decay != 0:
            g = g + weight_decay * p
        if k not in momentum_buf:
            momentum_buf[k] = g.clone()
        else:
            momentum_buf[k] = momentum_buf[k] * momentum + g
        g = g + momentum * momentum_buf[k]  # nesterov
        weights[k] = p - lr[grp] * g
    return float(total)


def ema_decay(updates: int, decay=0.9999) -> float:  # torch_utils.py:297
    return decay * (1 - math.exp(-updates / 2000))


def ema_update(ema: Dict[str, torch.Tensor], weights: Dict[str, torch.Tensor], updates: int) -> int:
    """utils/torch_utils.py:302-312: every floating-point state-dict entry (params AND BN running stats)."""
    updates += 1
    d = ema_decay(updates)
    for k, v in ema.items():
        if v.dtype.is_floating_point:
            ema[k] = v * d + (1.0 - d) * weights[k].detach()
    return updates


def warmup_lr(ni: int, nw: int, lr0: float, lf_epoch: float, warmup_bias_lr=0.0502, warmup_momentum=0.898, momentum=0.952):
    """trainers/base_trainer.py:100-112. Returns ([lr_g0, lr_g1, lr_g2], momentum).

    Quirk kept: optimizer.param_groups order is [g2 (bias), g0, g1] (averaging.py:264-266: the optimizer is
    created on g[2], then g[0], g[1] are added), and warmup treats group index j == 2 -- i.e. the BN-weight
    group -- as "bias" (SURVEY.md section 7 quirk d)."""
    xi = [0, nw]
    lrs_by_optimizer_index = [float(np.interp(ni, xi, [warmup_bias_lr if j == 2 else 0.0, lr0 * lf_epoch])) for j in range(3)]
    # optimizer index 0 -> g2 (biases), 1 -> g0 (decayed weights), 2 -> g1 (BN weights)
    lr = [lrs_by_optimizer_index[1], lrs_by_optimizer_index[2], lrs_by_optimizer_index[0]]
    mom = float(np.interp(ni, xi, [warmup_momentum, momentum]))
    return lr, mom
