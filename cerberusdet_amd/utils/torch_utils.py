"""Helpers kept from the reference's utils/torch_utils.py that the hot path's callers rely on:
get_hyperparameter (319-340), de_parallel (174-176), ModelEMA formula (282-316, fused into the optimizer kernel)."""
from __future__ import annotations

import math
from typing import Any, Dict, Optional


def get_hyperparameter(hyp: Dict[str, Any], name: str, task_ind: Optional[int] = None, task_name: Optional[str] = None):
    """Scalar, per-task list, or `{task}_{name}` / `{name}_{task}` keyed hyper-parameter."""
    if name not in hyp and task_name is not None:
        name = f"{task_name}_{name}" if f"{task_name}_{name}" in hyp else f"{name}_{task_name}"
    assert name in hyp, f"Requested not existed param {name}"
    param = hyp[name]
    if isinstance(param, list) and task_ind is not None:
        return param[task_ind]
    if isinstance(param, list):
        return param[0]
    return param


def de_parallel(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "blocks") else model


def ema_decay(updates: int, decay: float = 0.9999) -> float:
    """ModelEMA.decay (reference torch_utils.py:297)."""
    return decay * (1 - math.exp(-updates / 2000))
