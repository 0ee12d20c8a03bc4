#!/usr/bin/env python3
"""Build-container only: pins the PARAMETER / LABEL side of the training augmentation (cerberusdet_amd/augment.py) against the real
reference (`/root/reference`: data/datasets.py `LoadImagesAndLabels.__getitem__` / `load_mosaic`, data/augmentations.py
`random_perspective` / `augment_hsv` / `mixup`).

The reference's own functions run on a minimal stand-in for the dataset object (sizes, labels and cached images; no files) with the
global `random` / `np.random` generators seeded per sample. cv2 is absent from this image: a RECORDING stub captures what the reference
hands to it -- the matrix of every cv2.warpAffine / cv2.warpPerspective call, the three lookup tables of cv2.LUT -- and returns blank images; the one
value it computes, cv2.getRotationMatrix2D(center=(0, 0)), follows OpenCV's documented formula. Pixels are therefore NOT part of this
fixture (oracle/augment.py restates them, parity unpinned); what IS pinned: the order of the random draws, mosaic centre / partners /
paste rectangles (through the labels they move), M, the label warp + clip + box_candidates filter, mixup's partner and ratio, the HSV
tables, the flips and the final normalised labels of the batch dict.

Writes tests/golden/augment.json. Usage: python tools/make_golden_aug.py
"""
import json
import math
import random
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT / "tests" / "golden"))
sys.path.insert(0, str(ROOT / "tools"))
import synth  # noqa: E402

REC = {}


def _install_cv2():
    import make_golden

    make_golden._install_stubs()
    cv2 = sys.modules["cv2"]

    def getRotationMatrix2D(angle, center, scale):
        assert tuple(center) == (0, 0)
        a = math.radians(angle)
        al, be = scale * math.cos(a), scale * math.sin(a)
        return np.array([[al, be, 0.0], [-be, al, 0.0]])

    def warpAffine(im, M, dsize, borderValue=None):
        REC.setdefault("M", []).append(np.array(M, np.float64).tolist())
        assert tuple(borderValue) == (114, 114, 114)
        return np.zeros((dsize[1], dsize[0], 3), np.uint8)

    def warpPerspective(im, M, dsize, borderValue=None):
        REC.setdefault("M", []).append(np.array(M, np.float64).tolist())  # the full 3x3 matrix
        assert tuple(borderValue) == (114, 114, 114) and np.array(M).shape == (3, 3)
        return np.zeros((dsize[1], dsize[0], 3), np.uint8)

    def LUT(ch, lut):
        REC.setdefault("lut", []).append(np.asarray(lut).astype(int).tolist())
        return ch

    def copyMakeBorder(im, top, bottom, left, right, kind, value=None):
        REC.setdefault("border", []).append([int(top), int(bottom), int(left), int(right)])
        return np.zeros((im.shape[0] + top + bottom, im.shape[1] + left + right, 3), np.uint8)

    def resize(im, dsize, interpolation=None):
        REC.setdefault("resize", []).append([int(dsize[0]), int(dsize[1])])
        return np.zeros((dsize[1], dsize[0], 3), np.uint8)

    cv2.copyMakeBorder, cv2.resize = copyMakeBorder, resize
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR, cv2.INTER_AREA = 0, 1, 3
    cv2.getRotationMatrix2D = getRotationMatrix2D
    cv2.warpAffine = warpAffine
    cv2.warpPerspective = warpPerspective
    cv2.LUT = LUT
    cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 40, 54
    cv2.cvtColor = lambda im, code, dst=None: im
    cv2.split = lambda im: (im[..., 0], im[..., 1], im[..., 2])
    cv2.merge = lambda chs: np.stack(chs, -1)


def main():
    if not REF.exists():
        sys.exit("make_golden_aug.py needs /root/reference (build container only)")
    _install_cv2()
    sys.path.insert(0, str(REF))
    from cerberusdet.data.datasets import LoadImagesAndLabels

    cases = {}
    for name, c in synth.AUG_CASES.items():
        sizes, labels = synth.aug_dataset(c["seed"], c["n"], c["s"])
        fake = types.SimpleNamespace(
            img_size=c["s"], mosaic_border=[-c["s"] // 2, -c["s"] // 2], indices=range(c["n"]), n=c["n"], hyp=dict(c["hyp"]), augment=True, mosaic=True,
            rect=False, labels=[lb.copy() for lb in labels], imgs=[np.zeros((*synth.aug_resized(hw, c["s"]), 3), np.uint8) for hw in sizes],
            img_hw0=list(sizes), img_hw=[synth.aug_resized(hw, c["s"]) for hw in sizes], img_files=[f"img{i}" for i in range(c["n"])],
            albumentations=lambda im, lb: (im, lb))
        samples = []
        for k in range(c["samples"]):
            random.seed(c["seed"] * 1000 + k)
            np.random.seed(c["seed"] * 1000 + k)
            REC.clear()
            index = (7 * k + 3) % c["n"]
            img, labels_out, path, shapes = LoadImagesAndLabels.__getitem__(fake, index)
            assert tuple(img.shape) == (3, c["s"], c["s"])
            samples.append(dict(index=index, M=REC.get("M", []), lut=REC.get("lut"), labels=labels_out[:, 1:].numpy().astype(np.float64).tolist(),
                                shapes=None if shapes is None else [list(shapes[0]), [list(shapes[1][0]), list(shapes[1][1])]], border=REC.get("border"),
                                draws_after=[random.random(), float(np.random.uniform())]))
        cases[name] = samples
        print(name, [len(sm["labels"]) for sm in samples], [len(sm["M"]) for sm in samples])
    json.dump(cases, open(OUT / "augment.json", "w"))
    print("augment.json", (OUT / "augment.json").stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
