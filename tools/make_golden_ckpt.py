#!/usr/bin/env python3
"""Golden for the YOLOv8 -> CerberusDet state-dict remap: runs the REAL reference's utils/ckpt_utils.py:dict_to_cerber +
intersect_dicts (build container only) on a synthetic YOLO-style state dict and stores WHICH source entry each destination key
received (tensors are filled with their source's ordinal, so only key names and small ints are stored).

    python tools/make_golden_ckpt.py        # writes tests/golden/ckpt_remap.json
"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402  (stubs + tiny configs)


def yolo_dict_for(model, extra_head_nc):
    """A YOLOv8-shaped state dict for the model's architecture: backbone keys lose `blocks.0.`, neck blocks use their YOLO layer
    index, ONE detect head at the YOLO index of the first head (with `extra_head_nc` classes), plus two keys that cannot be
    placed (unknown layer, wrong shape)."""
    sd = model.state_dict()
    first_head = min(model.heads.values())
    head_i = model.blocks[first_head].i
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        b = int(parts[1])
        if b == 0:
            out[".".join(parts[2:])] = v.shape
        elif b == first_head:
            shp = list(v.shape)
            if "cv3" in k and parts[-3] == "2":  # final class projection: nc of the YOLO checkpoint
                shp[0] = extra_head_nc
            out[f"model.{head_i}." + ".".join(parts[2:])] = tuple(shp)
        elif b not in model.heads.values():
            out[f"model.{model.blocks[b].i}." + ".".join(parts[2:])] = v.shape
    out["model.97.conv.weight"] = (4, 4, 1, 1)                 # layer the model does not have
    k_bad = next(k for k in out if k.startswith("model.1.") and k.endswith("conv.weight"))
    out[k_bad] = tuple(list(out[k_bad])[:-1] + [7])            # shape mismatch in the backbone
    return {k: torch.full(tuple(s), float(i)) for i, (k, s) in enumerate(out.items())}


def main():
    if not mg.REF.exists():
        sys.exit("needs /root/reference (build container only)")
    mg._install_stubs()
    sys.path.insert(0, str(mg.REF))
    from cerberusdet.models.cerberus import CerberusDet
    from cerberusdet.utils.ckpt_utils import dict_to_cerber, intersect_dicts

    res = {}
    for name, three, ncs in (("tiny2", False, [20, 19]), ("tiny3", True, [20, 19, 12])):
        cfg = mg.tiny_cfg(three)
        tasks = ["voc", "objects365_animals", "objects365_tableware"][:len(ncs)]
        model = CerberusDet(tasks, ncs, cfg=cfg, ch=3, verbose=False)
        yolo = yolo_dict_for(model, extra_head_nc=ncs[0])
        mapped = dict_to_cerber(yolo, model)
        final = intersect_dicts(mapped, model.state_dict(), exclude=["anchor"])
        keys = list(yolo.keys())
        res[name] = dict(tasks=tasks, nc=ncs, yolo_keys=keys, yolo_shapes=[list(v.shape) for v in yolo.values()],
                         mapped={k: int(v.flatten()[0]) for k, v in mapped.items()},
                         final={k: int(v.flatten()[0]) for k, v in final.items()})
        print(name, len(yolo), "yolo keys ->", len(mapped), "mapped,", len(final), "after intersect")
    json.dump(res, open(mg.OUT / "ckpt_remap.json", "w"))
    print((mg.OUT / "ckpt_remap.json").stat().st_size / 1024, "KiB")


if __name__ == "__main__":
    main()
