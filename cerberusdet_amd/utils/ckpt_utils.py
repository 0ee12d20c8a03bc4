"""Checkpoint interop: load a plain YOLOv8 state dict (keys `model.{layer}.…`) into an un-split CerberusDet
(keys `blocks.{block}.…`). Same behaviour as the reference's `utils/ckpt_utils.py:5-90` (`intersect_dicts`, `dict_to_cerber`),
used by `ModelsManager.from_ckpt` (utils/models_manager.py:155-174) and `convert_to_cerber.py`:

* backbone layers (YOLO indices below the first neck block's index) -> `blocks.0.model.{i}.…` (the backbone block keeps the
  YOLO `model.` prefix inside it);
* each neck layer -> the block whose original YOLO index (`block.i`) it is: `blocks.{b}.…`;
* the YOLO detect head (the layer that owns `.dfl`) -> EVERY task head of the CerberusDet model (heads are then skipped by
  `intersect_dicts` wherever nc differs);
* unknown layers and shape mismatches are dropped (the reference logs a warning; so do we).
"""
from __future__ import annotations

import logging
from typing import Dict

LOGGER = logging.getLogger(__name__)


def intersect_dicts(da: Dict, db: Dict, exclude=()) -> Dict:
    """Entries of `da` whose key is in `db` with the same shape and contains none of the `exclude` substrings."""
    out = {}
    for k, v in da.items():
        if k not in db or v.shape != db[k].shape:
            continue
        if any(x in k for x in exclude):
            continue
        out[k] = v
    return out


def dict_to_cerber(loaded_dict: Dict, model) -> Dict:
    """YOLO state dict -> state dict for `model` (a CerberusDet). Returns only the entries that could be placed."""
    target = model.state_dict()
    head_blocks = list(model.heads.values())
    # the YOLO layer number of the detect head: the one that carries the DFL conv
    yolo_head = None
    for k in loaded_dict:
        if ".dfl" in k:
            yolo_head = k.split(".")[1]
    # YOLO layer index -> CerberusDet block index (0 = backbone: every layer in front of the first neck block)
    block_of_layer = {i: 0 for i in range(model.blocks[1].i)}
    for b in range(1, len(model.blocks)):
        block_of_layer[model.blocks[b].i] = b

    out = {}
    for k, v in loaded_dict.items():
        parts = k.split(".")
        if yolo_head is not None and f"model.{yolo_head}." in k:
            tail = ".".join(parts[2:])
            for hb in head_blocks:  # copied unchecked, like the reference: intersect_dicts filters afterwards
                out[f"blocks.{hb}.{tail}"] = v
            continue
        layer = int(parts[1])
        if layer not in block_of_layer:
            LOGGER.warning("YOLO key has not been mapped: %s", k)
            continue
        b = block_of_layer[layer]
        new_key = f"blocks.0.{k}" if b == 0 else f"blocks.{b}." + ".".join(parts[2:])
        if new_key not in target:
            LOGGER.warning("key %s has not been found in the CerberusDet state dict", new_key)
            continue
        if target[new_key].shape != v.shape:
            LOGGER.warning("mismatched shapes for %s: loaded %s, model %s", new_key, tuple(v.shape), tuple(target[new_key].shape))
            continue
        out[new_key] = v
    return out
