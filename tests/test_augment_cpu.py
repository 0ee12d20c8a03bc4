"""Host side of the training augmentation (cerberusdet_amd/augment.py) against the real reference's own functions
(tests/golden/augment.json, written by tools/make_golden_aug.py: `LoadImagesAndLabels.__getitem__` / `load_mosaic` / `random_perspective` /
`mixup` / `augment_hsv` of /root/reference run with seeded generators and a recording cv2 stub), and the numpy pixel restatement's
self-consistency. No GPU."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

import synth
from cerberusdet_amd import augment as A

GOLD = json.load(open(Path(__file__).parent / "golden" / "augment.json"))


@pytest.mark.parametrize("name", list(synth.AUG_CASES))
def test_parameters_and_labels_equal_the_reference(name):
    """Same generator states -> the same affine matrices (bit for bit), lookup tables, final labels, and the same number of draws consumed
    (the next draw of either generator agrees): mosaic centre, partner images, paste rectangles, M = T @ S @ R @ P @ C, label warp + clip +
    box_candidates, mixup partner + beta ratio, HSV gains, both flips."""
    c = synth.AUG_CASES[name]
    sizes, labels = synth.aug_dataset(c["seed"], c["n"], c["s"])
    n_mix = 0
    for k, want in enumerate(GOLD[name]):
        rng, nprng = random.Random(c["seed"] * 1000 + k), np.random.RandomState(c["seed"] * 1000 + k)
        plan = A.sample_plan(rng, nprng, want["index"], range(c["n"]), sizes, labels, c["s"], c["hyp"])
        assert len(plan.mosaics) == len(want["M"]), (name, k)
        n_mix += len(plan.mosaics) - 1
        for mo, M in zip(plan.mosaics, want["M"]):
            assert np.array_equal(mo.M[:2], np.array(M)), (name, k)
            for t in mo.tiles:  # the paste rectangle and the source window have the same extent and stay inside both images
                (x1a, y1a, x2a, y2a), (x1b, y1b) = t.dst, t.src
                assert 0 <= x1a <= x2a <= 2 * c["s"] and 0 <= y1a <= y2a <= 2 * c["s"]
                assert 0 <= x1b and x1b + (x2a - x1a) <= t.hw[1] and 0 <= y1b and y1b + (y2a - y1a) <= t.hw[0]
        if want.get("shapes") is None:
            assert plan.mosaics[0].shapes is None and plan.mosaics[0].canvas == 2 * c["s"]
        else:  # the single-image branch: the reference's `shapes` entry, and letterbox's integer border = where the tile sits
            (h0, w0), ((rh, rw), (dw, dh)) = plan.mosaics[0].shapes
            assert [[h0, w0], [[rh, rw], [dw, dh]]] == want["shapes"] and plan.mosaics[0].canvas == c["s"] and len(plan.mosaics) == 1
            t = plan.mosaics[0].tiles[0]
            top, bottom, left, right = want["border"][0]
            assert t.dst == (left, top, c["s"] - right, c["s"] - bottom) and t.src == (0, 0)
        if want["lut"] is None:
            assert plan.hsv_lut is None
        else:
            assert np.array_equal(plan.hsv_lut, np.array(want["lut"], np.uint8)), (name, k)
        got, ref = plan.labels, np.array(want["labels"], np.float32).reshape(-1, 6)
        assert got.shape == ref.shape and np.array_equal(got, ref), (name, k, got.shape, ref.shape)
        assert [rng.random(), float(nprng.uniform())] == want["draws_after"], (name, k)
    assert (n_mix > 0) == (c["hyp"]["mixup"] > 0)


def test_warp_coefficients_invert_the_matrix():
    rng = random.Random(3)
    M, s, w, h = A.sample_affine(rng, (256, 256), dict(A.HYP_DEFAULT, degrees=20.0, shear=8.0), (-64, -64))
    a = A.warp_coefficients(M)
    inv = np.array([[a[0], a[1], a[2]], [a[3], a[4], a[5]], [0, 0, 1]])
    assert np.allclose(inv @ M, np.eye(3), atol=1e-9) and (w, h) == (128, 128)


def test_pixel_restatement_identity_cases():
    """oracle/augment.py: an identity warp returns the canvas, an identity HSV table returns the image up to the 8-bit HSV round trip (<= 2
    levels per channel on saturated colours is OpenCV's own loss; grey pixels are exact), flips flip."""
    from oracle import augment as OA

    rng = np.random.RandomState(0)
    canvas = rng.randint(0, 256, (40, 48, 3)).astype(np.uint8)
    assert np.array_equal(OA.warp_affine_u8(canvas, np.eye(3), (48, 40)), canvas)
    shifted = OA.warp_affine_u8(canvas, np.array([[1, 0, 5.0], [0, 1, -3.0], [0, 0, 1]]), (48, 40))
    assert np.array_equal(shifted[0:37, 5:48], canvas[3:40, 0:43]) and (shifted[:, :5] == 114).all() and (shifted[37:] == 114).all()
    ident = np.stack([np.arange(256) % 180, np.arange(256), np.arange(256)]).astype(np.uint8)
    grey = np.repeat(rng.randint(0, 256, (8, 8, 1)), 3, 2).astype(np.uint8)
    assert np.array_equal(OA.augment_hsv(grey, ident), grey)
    back = OA.augment_hsv(canvas, ident).astype(int)
    assert np.abs(back - canvas.astype(int)).max() <= 4
    h, s, v = OA.bgr2hsv_u8(np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 255, 255]]], np.uint8))
    assert h.tolist() == [[120, 60, 0, 30]] and s.tolist() == [[255] * 4] and v.tolist() == [[255] * 4]  # pure blue / green / red / yellow
