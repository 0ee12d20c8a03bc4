"""Helpers kept from the reference's utils/torch_utils.py that the hot path's callers rely on:
get_hyperparameter (319-340), de_parallel (174-176), ModelEMA formula (282-316, fused into the optimizer kernel)."""
from __future__ import annotations

import math
from typing import Any, Dict, Optional


def parse_device(device: Any = "") -> int:
    """The GPU index a `--device` / `device=` value selects, in the reference's syntax (utils/torch_utils.py:75-100, select_device):
    '' -> 0, 'N' / 'cuda:N' -> N, 'N,M,...' -> N (the reference exports the list as CUDA_VISIBLE_DEVICES and trains on its first entry; here
    one process drives one GPU -- launch the others with torch.distributed.run). 'cpu' RAISES: the reference would train on the CPU
    (BASELINE config 1), this engine has no CPU path and must not silently move the run to a GPU. Touches no GPU state (safe before a launcher
    starts its ranks); `select_device` validates the index against the visible devices."""
    s = str(device if device is not None else "").strip().lower().replace("cuda:", "")
    if s == "cuda":
        s = ""
    if s == "cpu":
        raise RuntimeError("cerberusdet_amd: device='cpu' is not supported -- the engine runs on an MI355X only (the reference's device='cpu' path "
                           "is restated under oracle/ for tests and timed by bench.py's cpu_baseline)")
    if not s:
        return 0
    first = s.split(",")[0].strip()
    if not first.isdigit():
        raise ValueError(f"cerberusdet_amd: invalid device {device!r} (expected '', 'N', 'cuda:N' or 'N,M,...')")
    return int(first)


def select_device(device: Any = "", batch_size: Optional[int] = None):
    """Reference utils/torch_utils.py:75-100: the torch.device a run trains on. See parse_device for the accepted values."""
    import torch

    idx = parse_device(device)
    n = torch.cuda.device_count()
    if idx >= n:
        raise RuntimeError(f"cerberusdet_amd: device {device!r} requested but only {n} GPU(s) are visible")
    return torch.device("cuda", idx)


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
