#!/usr/bin/env python3
"""Golden fixture for the inference pre-processing (tests/golden/preprocess.json): the REAL reference's letterbox()
(cerberusdet/data/augmentations.py:59-89) and CerberusPreprocessor (cerberusdet_preprocessor.py) run with a RECORDING cv2 stub --
cv2 is absent here, so the pixel work of cv2.resize / cv2.copyMakeBorder cannot be executed; what IS the reference's own code, the
letterbox geometry (resized size, border split and rounding, check_img_size) and the call sequence, is recorded exactly.
Runs only in the build container (needs /root/reference)."""
import json
import os
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden"
assert REF.exists(), "needs /root/reference (build container only)"
sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden  # noqa: E402

make_golden._install_stubs()
cv2 = sys.modules["cv2"]
calls = []
cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = 1, 0


def _resize(im, dsize, interpolation=None):
    calls.append(("resize", list(dsize), interpolation))
    return np.zeros((dsize[1], dsize[0], 3), np.uint8)


def _border(im, top, bottom, left, right, btype, value=None):
    calls.append(("border", [top, bottom, left, right], list(value)))
    return np.zeros((im.shape[0] + top + bottom, im.shape[1] + left + right, 3), np.uint8)


cv2.resize, cv2.copyMakeBorder = _resize, _border
sys.path.insert(0, str(REF))
from cerberusdet.data.augmentations import letterbox  # noqa: E402

cases = []
for (h, w) in [(720, 1280), (1080, 1920), (480, 640), (640, 640), (375, 500), (1280, 720), (333, 777), (64, 48), (1281, 641), (2160, 3840)]:
    for img_size in (640, 416):
        for auto in (False, True):
            calls.clear()
            out, ratio, pad = letterbox(np.zeros((h, w, 3), np.uint8), img_size, stride=32, auto=auto)
            rs = [c for c in calls if c[0] == "resize"]
            bd = [c for c in calls if c[0] == "border"][0]
            cases.append(dict(h=h, w=w, img_size=img_size, auto=auto, resized=(rs[0][1] if rs else None), interpolation=(rs[0][2] if rs else None),
                              border=bd[1], color=bd[2], out_shape=list(out.shape[:2]), ratio=list(ratio), pad=list(pad)))
json.dump(dict(source="cerberusdet/data/augmentations.py:59-89 executed with a recording cv2 stub", INTER_LINEAR=1, cases=cases),
          open(OUT / "preprocess.json", "w"), indent=0)
print(len(cases), "cases ->", OUT / "preprocess.json")
