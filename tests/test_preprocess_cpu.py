"""Pre-processing oracle (oracle/preprocess.py) on the CPU: letterbox geometry against the real reference's recorded calls
(tests/golden/preprocess.json, tools/make_golden_pre.py), resize known answers of OpenCV's documented 8-bit bilinear, and the host
geometry of cerberusdet_amd.cerberusdet_preprocessor (which feeds the HIP kernel) against the same golden."""
import json
from pathlib import Path

import numpy as np

from oracle import preprocess as op

G = json.load(open(Path(__file__).resolve().parent / "golden" / "preprocess.json"))


def test_letterbox_geometry_matches_reference_calls():
    from cerberusdet_amd.cerberusdet_preprocessor import letterbox_geometry

    assert len(G["cases"]) == 40
    for c in G["cases"]:
        new_unpad, (top, bottom, left, right), ratio, pad = op.letterbox_geometry((c["h"], c["w"]), c["img_size"], auto=c["auto"], stride=32)
        if c["resized"] is None:
            assert (c["w"], c["h"]) == tuple(new_unpad)
        else:
            assert list(new_unpad) == c["resized"], c
        assert [top, bottom, left, right] == c["border"], c
        assert np.allclose(ratio, c["ratio"]) and np.allclose(pad, c["pad"])
        assert [new_unpad[1] + top + bottom, new_unpad[0] + left + right] == c["out_shape"]
        assert c["color"] == [114, 114, 114] and c["interpolation"] in (None, G["INTER_LINEAR"])
        # the product's host-side geometry (what the kernel is launched with)
        nw, nh, t, b, l, r = letterbox_geometry((c["h"], c["w"]), (c["img_size"], c["img_size"]), c["auto"], 32)
        assert [nw, nh] == list(new_unpad) and [t, b, l, r] == c["border"], c


def test_resize_known_answers():
    rng = np.random.default_rng(1)
    im = rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    assert np.array_equal(op.resize_linear_u8(im, (20, 12)), im)  # identity
    # exact 2x shrink = 2x2 area average, rounded
    want = ((im[0::2, 0::2].astype(int) + im[0::2, 1::2] + im[1::2, 0::2] + im[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(op.resize_linear_u8(im, (10, 6)), want)
    # 2x enlargement of a ramp: half-pixel centres -> 0.25 / 0.75 blends, borders replicate
    ramp = np.tile(np.arange(0, 80, 10, dtype=np.uint8)[None, :, None], (4, 1, 3))
    up = op.resize_linear_u8(ramp, (16, 8))
    assert up.shape == (8, 16, 3)
    # dst x: source coordinate (x + 0.5) / 2 - 0.5 -> -0.25 (clamped: 0), 0.25 (2.5 -> 3: the +2 >> 2 rounds half up), 0.75 (7.5 -> 8), ...
    assert up[0, :, 0].tolist() == [0, 3, 8, 13, 18, 23, 28, 33, 38, 43, 48, 53, 58, 63, 68, 70], up[0, :, 0].tolist()
    # constant image stays constant under any scale (coefficients sum to 2048)
    const = np.full((33, 47, 3), 137, np.uint8)
    assert (op.resize_linear_u8(const, (91, 18)) == 137).all()


def test_preprocess_layout_and_scaling():
    rng = np.random.default_rng(2)
    im = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    t = op.preprocess([im, im[::-1].copy()], 640, 32, half=False, auto=False)
    assert t.shape == (2, 3, 640, 640) and t.dtype == np.float32
    assert np.allclose(t[0, :, 0, 0], 114 / 255)  # top border (640x480 -> 80 rows of padding above and below)
    assert t[0, 0, 80, 0] == np.float32(im[0, 0, 2]) / np.float32(255)  # RGB <- BGR, no resize needed at this size
    assert t[0, 2, 80 + 479, 639] == np.float32(im[479, 639, 0]) / np.float32(255)
    th = op.preprocess([im], 640, 32, half=True)
    assert th.dtype == np.float16 and np.array_equal(th[0], t[0].astype(np.float16))
