"""Deterministic synthetic inputs shared by tools/make_golden.py (which feeds them to the real
reference) and by the tests (which feed the same values to the oracle / the HIP path).

Everything derives from numpy's MT19937 `RandomState`, whose streams are stable across numpy
versions and machines, so fixtures only need to store the reference's OUTPUTS.
"""
import zlib

import numpy as np


def _rng(seed, key=""):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def det_tensor(seed: int, key: str, shape) -> np.ndarray:
    """Value of state-dict entry `key` for fixture `seed` (reference key schema)."""
    rng = _rng(seed, key)
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, np.int64)
    if key.endswith("dfl.conv.weight"):
        return np.arange(16, dtype=np.float32).reshape(shape)
    if key.endswith("running_var"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith("running_mean"):
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if key.endswith("bn.weight"):
        return rng.uniform(0.7, 1.3, shape).astype(np.float32)
    if key.endswith("bn.bias"):
        return rng.uniform(-0.2, 0.2, shape).astype(np.float32)
    if key.endswith(".bias"):
        return rng.uniform(-1.0, 1.0, shape).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    b = (3.0 / fan_in) ** 0.5
    return rng.uniform(-b, b, shape).astype(np.float32)


def det_tensor_wc(seed: int, key: str, shape) -> np.ndarray:
    """Well-conditioned variant of det_tensor for the train-mode fixture `train_wc`: BatchNorm weight in [0.4, 0.6] and bias in [1, 2]
    keep the SiLU pre-activations in its near-linear range, so a random-weight BatchNorm network no longer amplifies a 16-bit storage
    rounding layer by layer (with det_tensor's values it does: head maps move by 5-12 % and gradient cosines fall to 0.3-0.8 when ONLY
    the weights are rounded to bf16, see test_train_fixture_sensitivity); the class-logit bias sits at -4 like a trained head
    (reference bias_init, models/yolo.py:102-110) so the BCE term is not dominated by 8400 confident false positives."""
    rng = _rng(seed, key)
    shape = tuple(shape)
    if key.endswith("bn.weight"):
        return rng.uniform(0.4, 0.6, shape).astype(np.float32)
    if key.endswith("bn.bias"):
        return rng.uniform(1.0, 2.0, shape).astype(np.float32)
    if ".cv3." in key and key.endswith(".2.bias"):
        return rng.uniform(-4.5, -3.5, shape).astype(np.float32)
    return det_tensor(seed, key, shape)


def sample(a, n=1024) -> np.ndarray:
    """Strided sample of at most ~n elements of an array (what train_wc.npz stores of every gradient / weight tensor)."""
    f = np.asarray(a).reshape(-1)
    return f[::max(1, f.size // n)].copy()


TRAIN_WC = dict(seed=21, bs=8, imgsz=128, boxes_per_img=3, iters=2)  # tests/golden/train_wc.npz (tools/make_golden.py train_wc)


def det_image(seed: int, bs: int, imgsz: int) -> np.ndarray:
    return _rng(seed, "image").uniform(0, 1, (bs, 3, imgsz, imgsz)).astype(np.float32)


def det_array(seed: int, key: str, shape, lo=-0.5, hi=0.5) -> np.ndarray:
    return _rng(seed, key).uniform(lo, hi, tuple(shape)).astype(np.float32)


def make_batch(bs, n_per_img, nc, seed, empty_images=()):
    """Label dict in the reference's collate format (data/datasets.py:440-459) as numpy arrays."""
    rng = _rng(seed, "labels")
    bi, cls, boxes = [], [], []
    for i in range(bs):
        if i in empty_images:
            continue
        for _ in range(n_per_img):
            bi.append(i)
            cls.append(rng.randint(0, nc))
            boxes.append(np.concatenate((rng.uniform(0.2, 0.8, 2), rng.uniform(0.05, 0.35, 2))))
    n = len(bi)
    return dict(
        batch_idx=np.asarray(bi, np.float32),
        cls=np.asarray(cls, np.float32).reshape(n, 1),
        prob=np.ones((n, 1), np.float32),
        bboxes=np.asarray(boxes, np.float32).reshape(n, 4),
    )


def synth_feats(seed, bs, imgsz, nc, mode="near"):
    """Three raw head maps [bs, 64+nc, h, w]. mode 'near' sharpens the DFL logits so that decoded
    boxes are a few cells wide (positive CIoU for many anchors); 'rand' is plain N(0,1)."""
    rng = _rng(seed, "feats")
    feats = []
    for s in (8, 16, 32):
        h = imgsz // s
        f = rng.normal(0, 1, (bs, 64 + nc, h, h)).astype(np.float32)
        if mode == "near":
            f[:, :64] *= 0.5
            v = f[:, :64].reshape(bs, 4, 16, h, h)
            v[:, :, 2:5] += 3.0
            f[:, :64] = v.reshape(bs, 64, h, h)
        feats.append(f)
    return feats


def synth_pred(bs, nc, na, n_obj, seed, dtype=np.float32, dup=True):
    """Synthetic eval-mode prediction y [bs, 4+nc, A] in the style of SURVEY.md section 8d."""
    rng = np.random.RandomState(seed)
    y = np.zeros((bs, 4 + nc, na), np.float32)
    y[:, 0:2] = rng.uniform(0, 640, (bs, 2, na))
    y[:, 2:4] = rng.uniform(10, 110, (bs, 2, na))
    y[:, 4:] = rng.uniform(0, 0.01, (bs, nc, na))
    for b in range(bs):
        if n_obj == 0:
            break
        idx = rng.choice(na, n_obj, replace=False)
        cls = rng.randint(0, nc, n_obj)
        y[b, 4 + cls, idx] = rng.uniform(0.25, 0.95, n_obj)
        if dup:  # clusters of near-duplicates so that suppression actually happens
            for k in range(0, n_obj - 1, 2):
                y[b, :4, idx[k + 1]] = y[b, :4, idx[k]] + rng.uniform(-3, 3, 4)
                y[b, 4:, idx[k + 1]] = 0.005
                y[b, 4 + cls[k], idx[k + 1]] = rng.uniform(0.25, 0.95)
    return y.astype(dtype)


# fixture case tables (single source of truth for generator and tests) ------------------------------
LOSS_CASES = {
    # name: (bs, imgsz, nc, labels/img, empty images, seed, pred mode)
    "basic": (2, 96, 20, 3, (), 11, "near"),
    "empty_image": (3, 64, 19, 2, (1,), 12, "near"),
    "no_labels": (2, 64, 12, 0, (0, 1), 13, "rand"),
    "random_preds": (2, 96, 20, 4, (), 14, "rand"),
    "many_gts": (2, 128, 20, 12, (), 15, "near"),
}

NMS_CASES = {
    "infer_fp32": dict(bs=3, nc=20, na=2100, n_obj=120, seed=7, dtype="float32", kw=dict(conf_thres=0.25, iou_thres=0.45)),
    "infer_fp16": dict(bs=2, nc=19, na=2100, n_obj=150, seed=8, dtype="float16", kw=dict(conf_thres=0.25, iou_thres=0.45)),
    "val_multilabel": dict(bs=2, nc=12, na=525, n_obj=60, seed=9, dtype="float32",
                           kw=dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)),
    "agnostic_maxdet": dict(bs=2, nc=20, na=2100, n_obj=400, seed=10, dtype="float32",
                            kw=dict(conf_thres=0.25, iou_thres=0.45, agnostic=True, max_det=50)),
    "classes_filter": dict(bs=2, nc=20, na=525, n_obj=80, seed=11, dtype="float32",
                           kw=dict(conf_thres=0.25, iou_thres=0.45, classes=[1, 3, 5])),
    "empty": dict(bs=2, nc=20, na=525, n_obj=0, seed=12, dtype="float32", kw=dict(conf_thres=0.25, iou_thres=0.45)),
}


def nms_case_input(name):
    c = NMS_CASES[name]
    return synth_pred(c["bs"], c["nc"], c["na"], c["n_obj"], c["seed"], np.dtype(c["dtype"]))


NMS_MASK_CASES = {
    # the reference's mask branch (general.py:410,443-449): nm coefficient channels behind the class scores ride along with the kept boxes
    "masks": dict(bs=2, nc=20, na=2100, n_obj=120, nm=4, seed=31, kw=dict(conf_thres=0.25, iou_thres=0.45, nm=4)),
    "masks_multilabel": dict(bs=2, nc=6, na=525, n_obj=60, nm=3, seed=32, kw=dict(conf_thres=0.05, iou_thres=0.5, multi_label=True, nm=3)),
    "masks_fp16_maxdet": dict(bs=1, nc=12, na=2100, n_obj=300, nm=8, seed=33, kw=dict(conf_thres=0.25, iou_thres=0.45, max_det=40, nm=8)),
}


def nms_mask_input(name):
    c = NMS_MASK_CASES[name]
    y = synth_pred(c["bs"], c["nc"], c["na"], c["n_obj"], c["seed"], np.dtype("float32"))
    m = np.random.RandomState(c["seed"] + 1000).uniform(-1, 1, (c["bs"], c["nm"], c["na"])).astype(np.float32)
    return np.concatenate((y, m), 1)


def ties_input():
    """Ties / touching boxes: IoU == thr must be kept (suppression is strict '>')."""
    y = np.zeros((1, 4 + 2, 8), np.float32)
    y[0, :4, 0] = [50, 50, 20, 20]
    y[0, :4, 1] = [60, 50, 20, 20]  # IoU with box0 = 1/3
    y[0, :4, 2] = [50, 50, 20, 20]  # exact duplicate of box0, same score -> index order decides
    y[0, :4, 3] = [200, 200, 30, 30]
    y[0, :4, 4] = [200, 200, 30, 30]  # duplicate but other class
    y[0, 4, [0, 1, 2, 3]] = [0.5, 0.5, 0.5, 0.9]
    y[0, 5, 4] = 0.9
    return y


def predict_inputs():
    """Per-task y for the CerberusDetInference.predict post-processing fixture."""
    ya = synth_pred(3, 20, 2100, 60, 21)
    yb = synth_pred(3, 19, 2100, 60, 22)
    for b in range(3):  # make some animals boxes coincide with voc boxes -> cross-task suppression triggers
        ia = np.nonzero(ya[b, 4:].max(0) > 0.25)[0][:10]
        ib = np.nonzero(yb[b, 4:].max(0) > 0.25)[0][:10]
        k = min(len(ia), len(ib))
        yb[b, :4, ib[:k]] = ya[b, :4, ia[:k]] + 0.5
    names = {"voc": [f"v{i}" for i in range(20)], "objects365_animals": [f"a{i}" for i in range(19)]}
    shapes = [(480, 640), (720, 1280), (640, 640)]
    return ya, yb, names, shapes


TINY_WIDTH, TINY_DEPTH = 0.125, 0.33
HYP = dict(box=[7.5, 7.5, 7.5], cls=[0.5, 0.5, 0.5], dfl=[1.5, 1.5, 1.5], lr0=0.00309, lrf=0.0956, momentum=0.952,
           weight_decay=0.00037, warmup_epochs=2.04, warmup_momentum=0.898, warmup_bias_lr=0.0502)


# ---------------------------------------------------------------------------------------------------------------------
# validation matcher / AP cases (tools/make_golden_val.py): (seed, predictions, labels, classes)
# ---------------------------------------------------------------------------------------------------------------------
VAL_CASES = [(11, 120, 25, 4), (12, 300, 60, 6), (13, 40, 0, 3), (14, 0, 7, 3), (15, 200, 150, 2), (16, 17, 3, 5)]


def val_case(seed, n, m, nc):
    """n predictions [n,6] (xyxy, conf desc, cls) and m labels [m,5] (cls, xyxy) in a 640 x 480 image; most predictions are
    jittered copies of labels (so IoUs spread over 0.3..1 and several predictions compete for one label), the rest is clutter."""
    rng = np.random.default_rng(seed)
    lab = np.zeros((m, 5), np.float32)
    if m:
        xy = rng.uniform(0, [540, 380], (m, 2))
        wh = rng.uniform(30, 100, (m, 2))
        lab[:, 1:3], lab[:, 3:5] = xy, xy + wh
        lab[:, 0] = rng.integers(0, nc, m)
    det = np.zeros((n, 6), np.float32)
    if n:
        for i in range(n):
            if m and rng.random() < 0.75:
                j = int(rng.integers(0, m))
                w, h = lab[j, 3] - lab[j, 1], lab[j, 4] - lab[j, 2]
                jit = rng.normal(0, 0.08, 4) * np.array([w, h, w, h])
                det[i, :4] = lab[j, 1:5] + jit
                det[i, 5] = lab[j, 0] if rng.random() < 0.85 else rng.integers(0, nc)
            else:
                xy = rng.uniform(0, [540, 380], 2)
                det[i, :2], det[i, 2:4] = xy, xy + rng.uniform(30, 100, 2)
                det[i, 5] = rng.integers(0, nc)
        det[:, 4] = np.sort(rng.uniform(0.01, 0.99, n).astype(np.float32))[::-1]
    return det, lab


# ---- training augmentation (tools/make_golden_aug.py, tests/test_augment_cpu.py, tests/test_gpu_augment.py) ------------------------
_AUG_HYP = dict(hsv_h=0.0124, hsv_s=0.696, hsv_v=0.287, degrees=0.299, translate=0.211, scale=0.846, scaleup=0.0, shear=0.717, perspective=0.0,
                flipud=0.00983, fliplr=0.5, mosaic=1.0, mixup=0.285)  # data/hyps/hyp.cerber-voc_obj365.yaml
AUG_CASES = {
    "hyp_default": dict(seed=41, n=12, s=640, samples=12, hyp=_AUG_HYP),
    "rotate_flip_mix": dict(seed=42, n=9, s=320, samples=10, hyp=dict(_AUG_HYP, degrees=10.0, shear=5.0, flipud=0.5, mixup=0.7, scaleup=0.6)),
    "half_mosaic": dict(seed=44, n=8, s=192, samples=10, hyp=dict(_AUG_HYP, mosaic=0.5, degrees=5.0)),  # both branches of __getitem__
    # hyp["perspective"] != 0: cv2.warpPerspective + the homogeneous division of the label corners (augmentations.py:152-153, 172)
    "perspective": dict(seed=45, n=8, s=256, samples=8, hyp=dict(_AUG_HYP, perspective=0.0008, degrees=8.0, mosaic=0.75, mixup=0.5)),
    "no_hsv_no_mix": dict(seed=43, n=6, s=256, samples=6, hyp=dict(_AUG_HYP, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, mixup=0.0, fliplr=0.0, flipud=0.0)),
}


def aug_resized(hw0, s):
    """reference load_image (data/datasets.py:470-477): long side -> s, truncated sizes."""
    h0, w0 = hw0
    r = s / max(h0, w0)
    return (int(h0 * r), int(w0 * r)) if r != 1 else (h0, w0)


def aug_dataset(seed, n, s):
    """n images of assorted sizes (some larger, some smaller than s, one exactly 2s on its long side, one equal to s) with 0..5 labels
    each: (sizes [(h0, w0)], labels [k, 6] float32 (cls, prob, x, y, w, h) normalised)."""
    rng = np.random.RandomState(seed)
    sizes, labels = [], []
    for i in range(n):
        if i == 0:
            hw = (2 * s, 2 * s - 2 * (s // 8))  # the exact 2x shrink in one direction only -> the bilinear path
        elif i == 1:
            hw = (2 * s, 2 * s)                 # exact 2x shrink: cv2's area fast path
        elif i == 2:
            hw = (s, s - s // 4)                # no resize
        else:
            hw = (int(rng.randint(s // 3, 2 * s)), int(rng.randint(s // 3, 2 * s)))
        sizes.append(hw)
        k = int(rng.randint(0, 6))
        cx, cy = rng.uniform(0.15, 0.85, k), rng.uniform(0.15, 0.85, k)
        w, h = rng.uniform(0.05, 0.28, k), rng.uniform(0.05, 0.28, k)
        lb = np.stack((rng.randint(0, 20, k).astype(np.float64), np.ones(k), cx, cy, w, h), 1).astype(np.float32)
        labels.append(lb.reshape(-1, 6))
    return sizes, labels


def aug_images(seed, sizes):
    """uint8 HWC BGR images with structure (gradients + blocks + noise) for the pixel tests."""
    out = []
    for i, (h, w) in enumerate(sizes):
        rng = np.random.RandomState(seed * 100 + i)
        yy, xx = np.mgrid[0:h, 0:w]
        im = np.stack(((xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx // 16 + yy // 16) % 2) * 200 + 20), -1).astype(np.int64)
        im = (im + rng.randint(-25, 26, im.shape)).clip(0, 255).astype(np.uint8)
        out.append(np.ascontiguousarray(im))
    return out
