"""Training-time augmentation for `train.py --data <yaml>` (default; `--no-augment` turns it off): mosaic + random affine + mixup + HSV + flips.

What the reference does per sample on CPU workers with cv2 (data/datasets.py:361-438 `__getitem__`, 483-542 `load_mosaic`;
data/augmentations.py:43-57 `augment_hsv`, 93-202 `random_perspective`, 205-211 `mixup`, 214-220 `box_candidates`), split in two here:

* HOST (this file, numpy): everything that consumes random numbers or touches labels, in the reference's order of draws -- mosaic centre
  and partner images, the affine matrix M = T @ S @ R @ P @ C, the label warp / clip / `box_candidates` filter, the mixup partner and its
  beta(32, 32) ratio, the HSV gains and their three 256-entry lookup tables, the two flips. The result is an `AugPlan`: a description of
  how every output pixel is produced. Given equal generator states it reproduces the reference's parameters and labels
  (`tests/golden/augment.json`, written by running the reference's own functions with a recording cv2 stub).
* GPU (csrc/augment.hip, `cdet_mosaic_augment_batch`): ONE kernel per batch produces the uint8 RGB NCHW images from the ORIGINAL decoded
  images: per output pixel undo the flips, map through the inverse affine (cv2.warpAffine's fixed-point bilinear, border 114), read the four
  taps from the virtual 2s x 2s mosaic canvas -- each tap is itself a bilinear sample of the original image (cv2.resize of `load_image`) --
  blend the mixup partner, apply BGR->HSV, the lookup tables, HSV->BGR, swap to RGB. No intermediate image is materialised.

Pixel arithmetic restates OpenCV's published 8-bit algorithms (no cv2 in this image): parity unpinned, like the letterbox kernel; the
kernel is bit-exact against `oracle/augment.py`, the numpy restatement of the same arithmetic.

Not reproduced: Albumentations (a no-op in the reference when the package is missing, augmentations.py:16-40), rectangular training,
image weights, the label cache. `perspective` != 0 (cold: every shipped hyper-parameter file uses 0.0) switches the warp to
cv2.warpPerspective's arithmetic and the label warp to the homogeneous division (augmentations.py:152-153, 172).
Both branches of `__getitem__` are covered: the mosaic (probability hyp["mosaic"], with mixup) and the single letterboxed image.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

HYP_DEFAULT = dict(hsv_h=0.0124, hsv_s=0.696, hsv_v=0.287, degrees=0.299, translate=0.211, scale=0.846, scaleup=0.0, shear=0.717,
                   perspective=0.0, flipud=0.00983, fliplr=0.5, mosaic=1.0, mixup=0.285)  # data/hyps/hyp.cerber-voc_obj365.yaml


@dataclass
class Tile:
    """One image of a mosaic: original size, size after `load_image` (long side -> s), where it lands on the canvas and from where."""
    index: int
    hw0: Tuple[int, int]
    hw: Tuple[int, int]
    dst: Tuple[int, int, int, int]  # x1a, y1a, x2a, y2a on the 2s x 2s canvas
    src: Tuple[int, int]            # x1b, y1b in the resized image


@dataclass
class Mosaic:
    tiles: List[Tile]
    M: np.ndarray            # 3x3 forward matrix (canvas -> output), float64
    labels: np.ndarray       # [n, 6] (cls, prob, x1, y1, x2, y2) in output pixels, after the candidate filter
    canvas: int = 0          # side of the square the tiles are pasted on: 2s for a mosaic, s for the single letterboxed image
    shapes: Optional[tuple] = None  # single image: ((h0, w0), ((h / h0, w / w0), (dw, dh))) like the reference's `shapes`
    perspective: bool = False  # hyp["perspective"] != 0: cv2.warpPerspective instead of cv2.warpAffine (augmentations.py:152-155)


@dataclass
class AugPlan:
    mosaics: List[Mosaic]                     # one, or two when mixup fires
    mix_ratio: Optional[float]                # weight of mosaics[0]
    hsv_lut: Optional[np.ndarray]             # [3, 256] uint8 (hue, sat, val) or None when all gains are 0
    flipud: bool
    fliplr: bool
    labels: np.ndarray = field(default_factory=lambda: np.zeros((0, 6), np.float32))  # (cls, prob, x, y, w, h) normalised, final


def resized_hw(hw0, s):
    """load_image (datasets.py:470-477): long side -> s, both sizes truncated."""
    h0, w0 = hw0
    r = s / max(h0, w0)
    return (int(h0 * r), int(w0 * r)) if r != 1 else (h0, w0)


def xywhn2xyxy(x, w, h, padw=0, padh=0):  # utils/general.py (labels of a tile -> canvas pixels)
    y = np.copy(x)
    y[:, 0] = w * (x[:, 0] - x[:, 2] / 2) + padw
    y[:, 1] = h * (x[:, 1] - x[:, 3] / 2) + padh
    y[:, 2] = w * (x[:, 0] + x[:, 2] / 2) + padw
    y[:, 3] = h * (x[:, 1] + x[:, 3] / 2) + padh
    return y


def xyxy2xywhn(x, w, h, clip=True, eps=1e-3):
    x = np.copy(x)
    if clip:
        x[:, [0, 2]] = x[:, [0, 2]].clip(0, w - eps)
        x[:, [1, 3]] = x[:, [1, 3]].clip(0, h - eps)
    y = np.copy(x)
    y[:, 0] = ((x[:, 0] + x[:, 2]) / 2) / w
    y[:, 1] = ((x[:, 1] + x[:, 3]) / 2) / h
    y[:, 2] = (x[:, 2] - x[:, 0]) / w
    y[:, 3] = (x[:, 3] - x[:, 1]) / h
    return y


def rotation_matrix(angle_deg, scale):
    """cv2.getRotationMatrix2D(center=(0, 0)) (documented formula): [[a, b, 0], [-b, a, 0]], a = s cos, b = s sin."""
    a = math.radians(angle_deg)
    al, be = scale * math.cos(a), scale * math.sin(a)
    return np.array([[al, be, 0.0], [-be, al, 0.0]])


def sample_affine(rng: random.Random, shape_hw, hyp, border=(0, 0)):
    """The matrix of random_perspective (augmentations.py:107-151) with its draws in order; returns (M, scale, width, height)."""
    height = shape_hw[0] + border[0] * 2
    width = shape_hw[1] + border[1] * 2
    Cm = np.eye(3)
    Cm[0, 2] = -shape_hw[1] / 2
    Cm[1, 2] = -shape_hw[0] / 2
    P = np.eye(3)
    P[2, 0] = rng.uniform(-hyp["perspective"], hyp["perspective"])
    P[2, 1] = rng.uniform(-hyp["perspective"], hyp["perspective"])
    R = np.eye(3)
    a = rng.uniform(-hyp["degrees"], hyp["degrees"])
    if not hyp["scaleup"]:
        s = rng.uniform(1 - hyp["scale"], 1 + hyp["scale"])
    elif rng.random() < 0.5:
        s = rng.uniform(1 - hyp["scale"], 1 + hyp["scale"])
    else:
        s = rng.uniform(1.09, 1 + hyp["scaleup"])
    R[:2] = rotation_matrix(a, s)
    S = np.eye(3)
    S[0, 1] = math.tan(rng.uniform(-hyp["shear"], hyp["shear"]) * math.pi / 180)
    S[1, 0] = math.tan(rng.uniform(-hyp["shear"], hyp["shear"]) * math.pi / 180)
    T = np.eye(3)
    T[0, 2] = rng.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]) * width
    T[1, 2] = rng.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]) * height
    return T @ S @ R @ P @ Cm, s, width, height


def box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.1, eps=1e-16):
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def warp_labels(targets, M, s, width, height, perspective=False):
    """Label part of random_perspective (augmentations.py:164-190): corners through M (divided by the homogeneous coordinate when
    perspective != 0, line 172), axis-aligned hull, clip, candidate filter."""
    n = len(targets)
    if not n:
        return targets
    xy = np.ones((n * 4, 3))
    xy[:, :2] = targets[:, [2, 3, 4, 5, 2, 5, 4, 3]].reshape(n * 4, 2)
    xy = xy @ M.T
    xy = (xy[:, :2] / xy[:, 2:3] if perspective else xy[:, :2]).reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    new = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
    new[:, [0, 2]] = new[:, [0, 2]].clip(0, width)
    new[:, [1, 3]] = new[:, [1, 3]].clip(0, height)
    keep = box_candidates(targets[:, 2:6].T * s, new.T, area_thr=0.10)
    targets = targets[keep]
    targets[:, 2:6] = new[keep]
    return targets


def sample_mosaic(rng: random.Random, index, indices: Sequence[int], sizes, labels, s, hyp) -> Mosaic:
    """load_mosaic (datasets.py:483-542). `sizes[i]` = (h0, w0) of image i, `labels[i]` = [n, 6] (cls, prob, x, y, w, h) normalised."""
    border = (-s // 2, -s // 2)
    yc, xc = (int(rng.uniform(-b, 2 * s + b)) for b in border)
    picks = [index] + rng.choices(list(indices), k=3)
    tiles, lab4 = [], []
    for i, idx in enumerate(picks):
        h, w = resized_hw(sizes[idx], s)
        if i == 0:  # top left
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b = w - (x2a - x1a), h - (y2a - y1a)
        elif i == 1:  # top right
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b = 0, h - (y2a - y1a)
        elif i == 2:  # bottom left
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b = w - (x2a - x1a), 0
        else:  # bottom right
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b = 0, 0
        tiles.append(Tile(int(idx), tuple(int(v) for v in sizes[idx]), (h, w), (x1a, y1a, x2a, y2a), (x1b, y1b)))
        lb = np.array(labels[idx]).reshape(-1, 6).copy()  # the dataset's own dtype (float32 label files): the reference's arithmetic up to the warp
        if lb.size:
            lb[:, 2:] = xywhn2xyxy(lb[:, 2:], w, h, x1a - x1b, y1a - y1b)
        lab4.append(lb)
    lab4 = np.concatenate(lab4, 0)
    np.clip(lab4[:, 2:], 0, 2 * s, out=lab4[:, 2:])
    M, sc, width, height = sample_affine(rng, (2 * s, 2 * s), hyp, border)
    persp = bool(hyp["perspective"])
    return Mosaic(tiles, M, warp_labels(lab4, M, sc, width, height, persp), canvas=2 * s, perspective=persp)


def sample_single(rng: random.Random, index, sizes, labels, s, hyp) -> Mosaic:
    """The non-mosaic branch of __getitem__ (datasets.py:376-402): load_image, letterbox(auto=False, scaleup=True), random_perspective
    without border. (When load_image's truncation leaves the long side at s - 1 the reference resizes a second time inside letterbox;
    here the original is resized once to that final size -- the labels are the reference's either way.)"""
    h0, w0 = sizes[index]
    h, w = resized_hw((h0, w0), s)
    r = min(s / h, s / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = (s - new_w) / 2, (s - new_h) / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    tile = Tile(int(index), (int(h0), int(w0)), (new_h, new_w), (left, top, left + new_w, top + new_h), (0, 0))
    lb = np.array(labels[index]).reshape(-1, 6).copy()
    if lb.size:
        lb[:, 2:] = xywhn2xyxy(lb[:, 2:], r * w, r * h, padw=dw, padh=dh)
    M, sc, width, height = sample_affine(rng, (s, s), hyp)
    persp = bool(hyp["perspective"])
    return Mosaic([tile], M, warp_labels(lb, M, sc, width, height, persp), canvas=s, shapes=((h0, w0), ((h / h0, w / w0), (dw, dh))),
                  perspective=persp)


def hsv_luts(nprng: np.random.RandomState, hyp):
    """augment_hsv (augmentations.py:43-57): the three lookup tables (None when every gain is 0: the reference skips the conversion)."""
    hg, sg, vg = hyp["hsv_h"], hyp["hsv_s"], hyp["hsv_v"]
    if not (hg or sg or vg):
        return None
    r = nprng.uniform(-1, 1, 3) * [hg, sg, vg] + 1
    x = np.arange(0, 256, dtype=r.dtype)
    return np.stack((((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8)))


def sample_plan(rng: random.Random, nprng: np.random.RandomState, index, indices, sizes, labels, s, hyp) -> AugPlan:
    """One training sample of the augmenting dataset (datasets.py:364-418, both branches), draws in the reference's order."""
    n = len(sizes)
    ratio = None
    if rng.random() < hyp["mosaic"]:
        mosaics = [sample_mosaic(rng, index, indices, sizes, labels, s, hyp)]
        if rng.random() < hyp["mixup"]:
            mosaics.append(sample_mosaic(rng, rng.randint(0, n - 1), indices, sizes, labels, s, hyp))
            ratio = float(nprng.beta(32.0, 32.0))
    else:
        mosaics = [sample_single(rng, index, sizes, labels, s, hyp)]
    lab = np.concatenate([m.labels for m in mosaics], 0)
    if len(lab):
        lab[:, 2:6] = xyxy2xywhn(lab[:, 2:6], w=s, h=s, clip=True, eps=1e-3)
    lut = hsv_luts(nprng, hyp)
    flipud = rng.random() < hyp["flipud"]
    if flipud and len(lab):
        lab[:, 3] = 1 - lab[:, 3]
    fliplr = rng.random() < hyp["fliplr"]
    if fliplr and len(lab):
        lab[:, 2] = 1 - lab[:, 2]
    return AugPlan(mosaics, ratio, lut, flipud, fliplr, lab.astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------- device side
def warp_coefficients(M):
    """cv2.warpAffine's view of a FORWARD matrix: the inverse 2x3 map (dst -> src) as OpenCV computes it (invertAffineTransform in
    double), handed to the kernel as six doubles."""
    A = np.asarray(M, np.float64)[:2]
    D = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = A[1, 1] * D, A[0, 0] * D
    a12, a21 = -A[0, 1] * D, -A[1, 0] * D
    b1 = -a11 * A[0, 2] - a12 * A[1, 2]
    b2 = -a21 * A[0, 2] - a22 * A[1, 2]
    return np.array([a11, a12, b1, a21, a22, b2], np.float64)


def warp_coefficients_perspective(M):
    """cv2.warpPerspective's view of a FORWARD matrix: the row-major 3x3 inverse (dst -> src) as cv::invert's closed 3x3 form computes
    it in double (determinant by the first row, cofactors times 1/det), nine doubles for the kernel."""
    m = np.asarray(M, np.float64)
    d = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
         + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    if d == 0:
        return np.zeros(9, np.float64)  # cv::invert leaves a zero matrix behind a singular input
    d = 1.0 / d
    return np.array([(m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d, (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d, (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d,
                     (m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d, (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d, (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d,
                     (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d, (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d, (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d],
                    np.float64)


def render_batch(plans: Sequence[AugPlan], images, s, device):
    """Launch cdet_mosaic_augment_batch for a batch of plans. `images[i]` = uint8 HWC BGR tensor of image i ON THE DEVICE (only the ones
    the plans use need to be present). Returns uint8 [B, 3, s, s] RGB."""
    import ctypes as C

    import torch

    from . import _lib as L

    lib = L.load()
    B = len(plans)
    samples = (L.AugSample * B)()
    keep = []
    for b, p in enumerate(plans):
        sm = samples[b]
        sm.n_mosaic = len(p.mosaics)
        sm.canvas = p.mosaics[0].canvas
        sm.mix_ratio = p.mix_ratio if p.mix_ratio is not None else 1.0
        sm.flipud, sm.fliplr = int(p.flipud), int(p.fliplr)
        sm.use_hsv = int(p.hsv_lut is not None)
        if p.hsv_lut is not None:
            C.memmove(sm.lut, np.ascontiguousarray(p.hsv_lut, np.uint8).ctypes.data, 768)
        sm.perspective = int(p.mosaics[0].perspective)
        for m, mo in enumerate(p.mosaics):
            assert mo.perspective == p.mosaics[0].perspective
            co = warp_coefficients_perspective(mo.M) if mo.perspective else warp_coefficients(mo.M)
            for k in range(len(co)):
                sm.minv[m * 9 + k] = float(co[k])
            for t, tl in enumerate(mo.tiles):
                it = sm.tiles[m * 4 + t]
                img = images[tl.index]
                assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.is_contiguous() and tuple(img.shape[:2]) == tuple(tl.hw0)
                keep.append(img)
                it.img, it.h0, it.w0, it.pitch = img.data_ptr(), tl.hw0[0], tl.hw0[1], tl.hw0[1] * 3
                it.h, it.w = tl.hw
                it.x1a, it.y1a, it.x2a, it.y2a = tl.dst
                it.x1b, it.y1b = tl.src
    tab = torch.frombuffer(bytearray(bytes(samples)), dtype=torch.uint8).to(device)
    out = torch.empty((B, 3, s, s), dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream(device)
    L.check(lib.cdet_mosaic_augment_batch(tab.data_ptr(), B, out.data_ptr(), s, st.cuda_stream), "cdet_mosaic_augment_batch")
    for t in keep + [tab]:
        t.record_stream(st)
    return out
