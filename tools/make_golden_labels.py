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
    json.dump(out, open(OUT / "labels.json", "w"), indent=1)
    print({k: (None if v is None else len(v)) for k, v in out["txt_cases"].items()}, {k: len(v) for k, v in out["xml_cases"].items()})


if __name__ == "__main__":
    main()
