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
            assert np.array_equal(mo.M[:len(M)], np.array(M)), (name, k)  # 2x3 handed to cv2.warpAffine, 3x3 to cv2.warpPerspective
            assert mo.perspective == (len(M) == 3) == bool(c["hyp"]["perspective"])
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


def test_perspective_coefficients_and_restatement():
    """hyp["perspective"] != 0 (augmentations.py:152-153): the nine kernel coefficients are the inverse homography; the numpy restatement of
    cv2.warpPerspective returns the canvas under the identity, equals the restated cv2.warpAffine for a pure integer shift (every weight
    exact in both arithmetics) and follows a real homography to within one bilinear cell of the float evaluation."""
    from oracle import augment as OA

    rng = random.Random(5)
    M, s, w, h = A.sample_affine(rng, (256, 256), dict(A.HYP_DEFAULT, degrees=10.0, perspective=0.001), (-64, -64))
    assert M[2, 0] != 0 and M[2, 1] != 0
    inv = A.warp_coefficients_perspective(M).reshape(3, 3)
    assert np.allclose(inv @ M, np.eye(3), atol=1e-9) and np.array_equal(inv, OA.invert3x3(M))
    nrng = np.random.RandomState(1)
    canvas = nrng.randint(0, 256, (128, 160, 3)).astype(np.uint8)  # wider than one 64-column block of the walk
    assert np.array_equal(OA.warp_perspective_u8(canvas, np.eye(3), (160, 128)), canvas)
    shift = np.array([[1, 0, 7.0], [0, 1, -4.0], [0, 0, 1]])
    assert np.array_equal(OA.warp_perspective_u8(canvas, shift, (160, 128)), OA.warp_affine_u8(canvas, shift, (160, 128)))
    smooth = np.zeros((256, 256, 3), np.uint8)
    smooth[..., 0], smooth[..., 1] = np.arange(256)[None, :], np.arange(256)[:, None]  # channel 0 = x, channel 1 = y of the source
    out = OA.warp_perspective_u8(smooth, M, (w, h)).astype(float)
    ys, xs = np.mgrid[0:h, 0:w]
    src = inv @ np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    sx, sy = (src[0] / src[2]).reshape(h, w), (src[1] / src[2]).reshape(h, w)
    inside = (sx > 1) & (sx < 254) & (sy > 1) & (sy < 254)
    assert inside.mean() > 0.3 and np.abs(out[..., 0] - sx)[inside].max() <= 1.0 and np.abs(out[..., 1] - sy)[inside].max() <= 1.0
    # labels: the corners go through the homogeneous division
    t = np.array([[0, 1, 100.0, 110.0, 150.0, 160.0]])
    got = A.warp_labels(t.copy(), M, 1.0, w, h, perspective=True)
    c = np.array([[100, 110, 1], [150, 160, 1], [100, 160, 1], [150, 110, 1.0]]) @ M.T
    c = c[:, :2] / c[:, 2:3]
    want = [c[:, 0].min().clip(0, w), c[:, 1].min().clip(0, h), c[:, 0].max().clip(0, w), c[:, 1].max().clip(0, h)]
    assert len(got) == 1 and np.allclose(got[0, 2:], want)


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
