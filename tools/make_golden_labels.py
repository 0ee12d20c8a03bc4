#!/usr/bin/env python3
"""Build-container only: label-file known answers from the real reference's `verify_image_label` (data/datasets.py:621-690; XML through
`xml_jsonify` / `convert_to_lb`, 545-618) -> tests/golden/labels.json. The inputs (label file texts) are part of the fixture; the images the
reference verifies alongside are tiny PNGs written to a temporary directory. Usage: python tools/make_golden_labels.py"""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT / "tools"))

XML = """<annotation><folder>f</folder><filename>a.png</filename><path>p</path>
<size><width>200</width><height>100</height><depth>3</depth></size>
<object><name>cat</name><bndbox><xmin>20.7</xmin><ymin>10</ymin><xmax>120.2</xmax><ymax>90.9</ymax></bndbox>
 <minors><minor><name>dog</name><votes>1</votes></minor><minor><name>cat</name><votes>3</votes></minor></minors></object>
<object><name>bird</name><bndbox><xmin>150</xmin><ymin>5</ymin><xmax>190</xmax><ymax>45</ymax></bndbox>
 <minors><minor><name>dog</name><votes>2</votes></minor></minors></object>
<object><name>dog</name><bndbox><xmin>0</xmin><ymin>0</ymin><xmax>50</xmax><ymax>50</ymax></bndbox></object>
</annotation>"""
NAMES = ["bird", "cat", "dog"]
TXT = {
    "plain": "3 0.5 0.5 0.2 0.2\n1 0.25 0.25 0.1 0.1\n",
    "duplicates": "3 0.5 0.5 0.2 0.2\n1 0.25 0.25 0.1 0.1\n3 0.5 0.5 0.2 0.2\n0 0.1 0.1 0.1 0.1\n",
    "six_columns": "3 1.0 0.5 0.5 0.2 0.2\n",
    "mixed_columns": "3 0.5 0.5 0.2 0.2\n1 0.7 0.25 0.25 0.1 0.1\n",
    "negative": "3 0.5 -0.5 0.2 0.2\n",
    "out_of_bounds": "3 0.5 0.5 1.2 0.2\n",
    "empty": "",
}


def main():
    if not REF.exists():
        sys.exit("needs /root/reference (build container only)")
    import make_golden

    make_golden._install_stubs()
    sys.path.insert(0, str(REF))
    from PIL import Image

    from cerberusdet.data.datasets import verify_image_label

    out = dict(xml=XML, names=NAMES, txt=TXT, xml_cases={}, txt_cases={})
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        Image.fromarray(np.zeros((32, 32, 3), np.uint8)).save(td / "a.png")
        (td / "a.xml").write_text(XML)
        for multi in (False, True):
            for soft in (False, True):
                r = verify_image_label((str(td / "a.png"), str(td / "a.xml")), "", True, NAMES, multi, soft)
                out["xml_cases"][f"multi{int(multi)}_soft{int(soft)}"] = r[1].astype(np.float64).tolist()
        for name, text in TXT.items():
            (td / "t.txt").write_text(text)
            r = verify_image_label((str(td / "a.png"), str(td / "t.txt")), "", False, NAMES, False, False)
            out["txt_cases"][name] = None if r[1] is None else r[1].astype(np.float64).tolist()
    # the class-balanced sampler (data/samplers.py:9-101) on a small label table, seeded
    from cerberusdet.data.samplers import BalancedBatchSampler

    rng = np.random.RandomState(5)
    table = []
    for i in range(40):
        k = int(rng.randint(0, 4))
        cls = rng.choice([0, 0, 0, 1, 1, 2, 5, 7], k)
        table.append(np.concatenate((cls[:, None].astype(np.float32), np.ones((k, 5), np.float32)), 1).reshape(-1, 6))
    class _DS(list):  # len() + .indices + .labels
        indices, labels = range(len(table)), table

    ds = _DS(table)
    epochs = []
    for seed in (0, 1):
        np.random.seed(seed)
        epochs.append([int(i) for i in BalancedBatchSampler(ds)])
    json.dump(dict(table=[t[:, 0].astype(int).tolist() for t in table], epochs=epochs), open(OUT / "sampler.json", "w"))
    # rectangular validation batches (datasets.py:270-289, 376-407 with rect=True, pad=0.5, augment=False) on real files
    import random

    import cv2  # the stub of make_golden

    from cerberusdet.data.datasets import LoadImagesAndLabels

    rec = {}
    cv2.imread = lambda path: np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    cv2.resize = lambda im, dsize, interpolation=None: np.zeros((dsize[1], dsize[0], 3), np.uint8)
    cv2.copyMakeBorder = lambda im, t, b, l, r, kind, value=None: np.zeros((im.shape[0] + t + b, im.shape[1] + l + r, 3), np.uint8)
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR, cv2.INTER_AREA = 0, 1, 3
    sizes = [(375, 500), (500, 375), (480, 640), (333, 500), (500, 333), (640, 640), (300, 900), (900, 300), (427, 640), (360, 480), (97, 333)]
    rect = dict(sizes=sizes, imgsz=256, batch=4, items=[])
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "images").mkdir()
        (td / "labels").mkdir()
        rng = np.random.RandomState(3)
        labels = []
        for i, (h, w) in enumerate(sizes):
            Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(td / "images" / f"{i:02d}.png")
            k = int(rng.randint(0, 4))
            rows = [[int(rng.randint(0, 20)), *rng.uniform(0.2, 0.8, 2), *rng.uniform(0.05, 0.3, 2)] for _ in range(k)]
            labels.append(rows)
            (td / "labels" / f"{i:02d}.txt").write_text("".join(f"{r[0]} {r[1]:.6f} {r[2]:.6f} {r[3]:.6f} {r[4]:.6f}\n" for r in rows))
        rect["label_rows"] = labels
        ds = LoadImagesAndLabels(str(td / "images"), 256, 4, augment=False, hyp=None, rect=True, stride=32, pad=0.5, classnames=[str(i) for i in range(20)])
        rect["batch_shapes"] = ds.batch_shapes.tolist()
        rect["order"] = [int(Path(f).stem) for f in ds.img_files]
        for idx in range(len(ds)):
            img, lab, path, shapes = ds[idx]
            rect["items"].append(dict(img_hw=list(img.shape[1:]), labels=lab[:, 1:].numpy().astype(np.float64).tolist(),
                                      shapes=[list(shapes[0]), [list(shapes[1][0]), list(shapes[1][1])]]))
    json.dump(rect, open(OUT / "rect_val.json", "w"))
    json.dump(out, open(OUT / "labels.json", "w"), indent=1)
    print({k: (None if v is None else len(v)) for k, v in out["txt_cases"].items()}, {k: len(v) for k, v in out["xml_cases"].items()})


if __name__ == "__main__":
    main()
