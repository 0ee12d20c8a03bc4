"""Plain data path for `train.py --data <yaml>`: YOLO-txt labels -> the reference's batch dicts (data/datasets.py:440-459).

What the reference does per sample without augmentation (data/datasets.py:361-438 with augment=False / rect=False): load the image,
letterbox it to a square of `imgsz` (data/augmentations.py:59-89, auto=False), move the normalised xywh labels into the letterboxed
frame (xywhn2xyxy with ratio / pad, then xyxy2xywhn with clip, utils/general.py) and collate
{"img" uint8 [N,3,H,W] RGB, "cls" [n,1], "prob" [n,1], "bboxes" [n,4] xywh in [0,1], "batch_idx" [n], "im_file", "ori_shape", "ratio_pad"}.
Here the host only decodes the files (PIL) and parses the labels; resize + border + layout run as ONE HIP kernel per batch
(csrc/preprocess.hip, the same kernel CerberusPreprocessor uses, uint8 output) and the batch is born on the GPU.

Dataset YAML = the reference's (data/voc_obj365_animals.yaml): `train` / `val`: one image directory (or .txt list) per task,
`nc`, `names`, `task_ids`. Label files sit next to the images with /images/ replaced by /labels/ (datasets.py:90-103): txt rows
`cls x y w h` (datasets.py:654-666), or with --labels-from-xml the reference's annotation XML incl. the per-box votes of other classes
(--use-multi-labels / --use-soft-labels, datasets.py:545-618).

`augment=True` (train.py's default; --no-augment turns it off): the reference's training augmentation -- mosaic of four images, random affine, mixup, HSV, flips
(data/datasets.py:361-438, 483-542; data/augmentations.py:43-211) -- with every random draw and the label geometry on the host
(cerberusdet_amd/augment.py, pinned against the reference's own functions) and the pixels rendered by ONE kernel per batch straight from
the decoded originals (csrc/augment.hip). Training loaders draw their epoch with the reference's class-balanced sampler (`balanced_order`).
Validation loaders are rectangular like the reference's (`rect_batch_shapes`); non-augmented images shrink with cv2.INTER_AREA's
arithmetic like `load_image` does. Not reproduced: the label cache, Albumentations, torch's own randperm inside DistributedSampler (a numpy permutation
shards the balanced draw over the ranks).
Shards: rank r takes samples r, r + world, ... of the (per-epoch, seeded) permutation, like a DistributedSampler.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


IMG_EXT = {".bmp", ".jpg", ".jpeg", ".png", ".tif", ".tiff", ".webp"}


def img2label_path(image_path: str, use_xml: bool = False) -> str:
    """reference data/datasets.py:90-103: /images/ -> /labels/, .txt or (--labels-from-xml) .xml."""
    sa, sb = os.sep + "images" + os.sep, os.sep + "labels" + os.sep
    return sb.join(str(Path(image_path).with_suffix(".xml" if use_xml else ".txt")).rsplit(sa, 1))


def list_images(path: str) -> List[str]:
    p = Path(path)
    if p.is_dir():
        files = [str(f) for f in sorted(p.rglob("*")) if f.suffix.lower() in IMG_EXT]
    elif p.is_file():  # a text file of image paths (relative ones resolve against its directory)
        files = [str((p.parent / ln[2:]) if ln.startswith("./") else ln) for ln in p.read_text().strip().splitlines() if ln.strip()]
    else:
        raise FileNotFoundError(f"{path} does not exist")
    if not files:
        raise FileNotFoundError(f"no images found under {path}")
    return files


def xml_annotation(label_path: str) -> dict:
    """The reference's annotation XML (datasets.py:545-586 xml_jsonify): size, and per <object> its name, integer-truncated box and the
    optional <minors> votes of other classes."""
    import xml.etree.ElementTree as ET

    root = ET.parse(label_path).getroot()
    ann = dict(width=int(root.find("size").find("width").text), height=int(root.find("size").find("height").text), bounding_boxes=[])
    for obj in root.findall("object"):
        bbox, minors = obj.find("bndbox"), obj.find("minors")
        ann["bounding_boxes"].append(dict(
            cls=obj.find("name").text, x_min=int(float(bbox.find("xmin").text)), y_min=int(float(bbox.find("ymin").text)),
            x_max=int(float(bbox.find("xmax").text)), y_max=int(float(bbox.find("ymax").text)),
            # (`if minors` in the reference: an Element without children is falsy)
            minors={x.find("name").text: int(x.find("votes").text) for x in minors} if minors is not None and len(minors) else None))
    return ann


def labels_from_annotation(ann: dict, classnames: Sequence[str], as_multi_label: bool, as_soft_label: bool) -> np.ndarray:
    """datasets.py:589-618 convert_to_lb: one row per (box, class) -- the main class, plus the minor classes with --use-multi-labels;
    prob = share of the annotators' votes with --use-soft-labels, else 1."""
    lb = []
    for bb in ann["bounding_boxes"]:
        cx = (bb["x_max"] + bb["x_min"]) / 2 / ann["width"]
        cy = (bb["y_max"] + bb["y_min"]) / 2 / ann["height"]
        w = (bb["x_max"] - bb["x_min"]) / ann["width"]
        h = (bb["y_max"] - bb["y_min"]) / ann["height"]
        votes = dict(bb["minors"]) if bb["minors"] else {}
        if bb["cls"] not in votes:  # the main class without a vote count of its own: one more than all minors together
            votes[bb["cls"]] = sum(votes.values()) + 1
        if as_soft_label:
            tot = sum(votes.values())
            votes = {k: v / tot for k, v in votes.items()}
        else:
            votes = {k: 1 for k in votes}
        if not as_multi_label:
            votes = {k: v for k, v in votes.items() if k == bb["cls"]}
        for c, p in votes.items():
            lb.append([classnames.index(c), p, cx, cy, w, h])
    return np.array(lb, dtype=np.float32)


def read_labels(label_path: str, nc: int, use_xml: bool = False, classnames: Optional[Sequence[str]] = None, as_multi_label: bool = False,
                as_soft_label: bool = False) -> np.ndarray:
    """-> [n, 6] float32 (cls, prob, x, y, w, h); missing file = background image. The label half of verify_image_label
    (datasets.py:621-690): txt rows are `cls x y w h` (a probability column of 1.0 is inserted; rows with more columns are an "Invalid
    annotation file"), XML files go through xml_annotation / labels_from_annotation; values must be >= 0 and coordinates <= 1; duplicate rows
    are removed the way the reference does it (np.unique's sorted order, only when there ARE duplicates)."""
    if not os.path.isfile(label_path):
        return np.zeros((0, 6), np.float32)
    if use_xml:
        assert classnames is not None, "XML labels name their classes: the dataset YAML needs `names`"
        lb = labels_from_annotation(xml_annotation(label_path), list(classnames), as_multi_label, as_soft_label)
    else:
        rows = [ln.split() for ln in Path(label_path).read_text().strip().splitlines() if len(ln)]
        if any(len(r) == 5 for r in rows):
            rows = [[r[0], "1.0", *r[1:]] for r in rows]
        elif any(len(r) > 5 for r in rows):
            raise ValueError("Invalid annotation file")
        lb = np.array(rows, dtype=np.float32)
    if not len(lb):
        return np.zeros((0, 6), np.float32)
    assert lb.ndim == 2 and lb.shape[1] == 6, "labels require 6 columns each"
    assert (lb >= 0).all(), "negative labels"
    assert (lb[:, 2:] <= 1).all(), "non-normalized or out of bounds coordinate labels"
    assert lb[:, 0].max() < nc, f"class {int(lb[:, 0].max())} exceeds nc = {nc}"
    _, i = np.unique(lb, axis=0, return_index=True)
    if len(i) < len(lb):
        lb = lb[i]
    return lb


def letterbox_labels(lb_xywhn: np.ndarray, shape_hw, imgsz: int, frame_hw=None, scaleup: bool = False):
    """Normalised xywh of the original image -> normalised xywh of its letterboxed frame (`imgsz` x `imgsz`, or the rectangular
    `frame_hw` of its validation batch), plus the reference's `shapes` entry ((h0, w0), ((h / h0, w / w0), (dw, dh))) that val uses to map
    boxes back (datasets.py:376-407, 428): load_image brings the long side to imgsz (truncated sizes), letterbox(auto=False,
    scaleup=augment) pads -- and shrinks once more when the frame is smaller than the resized image."""
    h0, w0 = shape_hw
    H, W = (imgsz, imgsz) if frame_hw is None else frame_hw
    r = imgsz / max(h0, w0)
    h, w = (int(h0 * r), int(w0 * r)) if r != 1 else (h0, w0)
    r2 = min(H / h, W / w)
    if not scaleup:
        r2 = min(r2, 1.0)
    new_w, new_h = int(round(w * r2)), int(round(h * r2))
    dw, dh = (W - new_w) / 2, (H - new_h) / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    out = lb_xywhn.copy()
    if len(out):
        x, y, bw, bh = (lb_xywhn[:, i] for i in range(4))
        x1, y1 = r2 * w * (x - bw / 2) + dw, r2 * h * (y - bh / 2) + dh
        x2, y2 = r2 * w * (x + bw / 2) + dw, r2 * h * (y + bh / 2) + dh
        eps = 1e-3
        x1, x2 = np.clip(x1, 0, W - eps), np.clip(x2, 0, W - eps)
        y1, y2 = np.clip(y1, 0, H - eps), np.clip(y2, 0, H - eps)
        out = np.stack(((x1 + x2) / 2 / W, (y1 + y2) / 2 / H, (x2 - x1) / W, (y2 - y1) / H), 1).astype(np.float32)
    return out, ((h0, w0), ((h / h0, w / w0), (dw, dh))), (new_w, new_h, top, left)


def rect_batch_shapes(sizes_hw: Sequence, batch_size: int, imgsz: int, stride: int = 32, pad: float = 0.5):
    """Rectangular batches of the reference's validation loaders (datasets.py:270-289 with rect=True, pad=0.5; utils/train_utils.py:45-57):
    images sorted by aspect ratio h / w, every batch gets the smallest stride-multiple frame that holds its images at long side imgsz.
    -> (order [n], frames [(H, W)] per batch)."""
    hw = np.array(sizes_hw, dtype=np.float64).reshape(-1, 2)
    ar = hw[:, 0] / hw[:, 1]
    order = ar.argsort()
    ar = ar[order]
    n = len(ar)
    bi = np.floor(np.arange(n) / batch_size).astype(int)
    nb = bi[-1] + 1
    shapes = [[1, 1]] * nb
    for i in range(nb):
        ari = ar[bi == i]
        mini, maxi = ari.min(), ari.max()
        if maxi < 1:
            shapes[i] = [maxi, 1]
        elif mini > 1:
            shapes[i] = [1, 1 / mini]
    frames = np.ceil(np.array(shapes) * imgsz / stride + pad).astype(int) * stride
    return order, [tuple(int(v) for v in f) for f in frames]


def balanced_order(labels: Sequence[np.ndarray], nprng: np.random.RandomState) -> np.ndarray:
    """One epoch of the reference's BalancedBatchSampler(class_choice="least_sampled") (data/samplers.py:9-101; train.py:95 turns it on for
    every training loader): len(dataset) draws; each picks the class with the fewest sampled labels so far -- ties at random, the first class
    of the table twice as likely as the others, as the reference's tie list is built -- then an image of that class at random, and counts
    all labels of the chosen image. Images without labels are never drawn. Same draws as the reference for the same generator state
    (tests/golden/sampler.json)."""
    per_image = [list(map(int, lb[:, 0].tolist())) for lb in labels]
    class_indices: Dict[int, List[int]] = {}
    for idx, cls in enumerate(per_image):
        for c in cls:
            class_indices.setdefault(c, []).append(idx)
    if not class_indices:
        return np.arange(len(labels))
    classes = list(class_indices)
    counts = {c: 0 for c in classes}
    out = []
    for _ in range(len(labels)):
        first = classes[0]
        lo, ties = counts[first], [first]
        for c in classes:
            if counts[c] < lo:
                lo, ties = counts[c], [c]
            if counts[c] == lo:
                ties.append(c)
        cls = ties[nprng.randint(0, len(ties))]
        pool = class_indices[cls]
        pick = pool[nprng.randint(0, len(pool))]
        for c in per_image[pick]:
            counts[c] += 1
        out.append(pick)
    return np.array(out, np.int64)


class TaskDataset:
    """Iterable over one task's batches for one rank; `len()` = batches per epoch. Every `iter()` starts a new epoch."""

    def __init__(self, path: str, imgsz: int, batch_size: int, nc: int, device, rank: int = 0, world_size: int = 1, shuffle: bool = True,
                 seed: int = 0, augment: bool = False, hyp: Optional[dict] = None, labels_from_xml: bool = False, classnames=None,
                 use_multi_labels: bool = False, use_soft_labels: bool = False, balanced: bool = False, rect: bool = False, stride: int = 32,
                 pad: float = 0.5, single_cls: bool = False):
        self.balanced, self.rect, self.stride, self.pad = balanced, rect, stride, pad
        assert not (rect and (augment or shuffle or balanced or world_size > 1)), "rectangular batches are the (unsharded, ordered) validation form"
        self.files, self.labels = [], []
        for f in list_images(path):  # like verify_image_label: an image whose label file does not verify is left out with a warning
            try:
                lb = read_labels(img2label_path(f, labels_from_xml), nc, labels_from_xml, classnames, use_multi_labels, use_soft_labels)
            except Exception as e:  # noqa: BLE001
                print(f"WARNING: Ignoring corrupted image and/or label {f}: {e}")
                continue
            if single_cls:  # datasets.py:258-260
                lb[:, 0] = 0
            self.files.append(f)
            self.labels.append(lb)
        if not self.files:
            raise FileNotFoundError(f"no usable image / label pair under {path}")
        self.imgsz, self.bs, self.nc, self.device = imgsz, batch_size, nc, torch.device(device)
        self.rank, self.world, self.shuffle, self.seed, self.epoch = rank, world_size, shuffle, seed, 0
        self.augment = augment
        if augment or rect:
            from PIL import Image

            self.sizes = []
            for f in self.files:  # header reads only: mosaic geometry / batch frames need every image's size before any pixel is decoded
                with Image.open(f) as im:
                    self.sizes.append((im.size[1], im.size[0]))
        if augment:
            from .augment import HYP_DEFAULT

            self.hyp = dict(HYP_DEFAULT, **{k: v for k, v in (hyp or {}).items() if k in HYP_DEFAULT})
        self.frames = None
        if rect:
            order, self.frames = rect_batch_shapes(self.sizes, batch_size, imgsz, stride, pad)
            self.files = [self.files[i] for i in order]
            self.labels = [self.labels[i] for i in order]
            self.sizes = [self.sizes[i] for i in order]
        per_rank = (len(self.files) + world_size - 1) // world_size
        self.nb = max((per_rank + batch_size - 1) // batch_size, 1)

    def __len__(self):
        return self.nb

    def _order(self):
        n = len(self.files)
        if self.balanced:  # the class-balanced draw of the epoch, then (like DistributedSamplerWrapper) a shuffle of ITS positions over the ranks
            rs = np.random.RandomState(self.seed * 7919 + self.epoch)
            idx = balanced_order(self.labels, rs)
            if self.world > 1:
                idx = idx[rs.permutation(len(idx))]
        else:
            idx = np.random.RandomState(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        pad = (-n) % self.world  # pad by wrapping so that every rank sees the same number of samples
        idx = np.concatenate((idx, idx[:pad])) if pad else idx
        return idx[self.rank::self.world]

    def _iter_augmented(self, order):
        """Mosaic / affine / mixup / HSV / flip batches (the reference's `__getitem__` with augment=True + `collate_fn`)."""
        import random

        from PIL import Image

        from . import augment as A

        S, n = self.imgsz, len(self.files)
        key = (self.seed * 1000003 + self.epoch) * 64 + self.rank
        rng, nprng = random.Random(key), np.random.RandomState(key % (2 ** 32))
        for b0 in range(0, len(order), self.bs):
            ids = order[b0:b0 + self.bs]
            plans = [A.sample_plan(rng, nprng, int(i), range(n), self.sizes, self.labels, S, self.hyp) for i in ids]
            need = sorted({t.index for p in plans for m in p.mosaics for t in m.tiles})
            images = {}
            for i in need:  # cv2.imread order: BGR
                images[i] = torch.from_numpy(np.ascontiguousarray(np.asarray(Image.open(self.files[i]).convert("RGB"))[:, :, ::-1])).to(self.device, non_blocking=True)
            img = A.render_batch(plans, images, S, self.device)
            lab = [p.labels for p in plans]
            cat = lambda col, w: torch.from_numpy(np.concatenate([lb[:, col] for lb in lab], 0).reshape(-1, w).astype(np.float32)).to(self.device)  # noqa: E731
            bidx = np.concatenate([np.full(len(lb), j, np.float32) for j, lb in enumerate(lab)], 0)
            yield {"img": img, "cls": cat(slice(0, 1), 1), "prob": cat(slice(1, 2), 1), "bboxes": cat(slice(2, 6), 4),
                   "batch_idx": torch.from_numpy(bidx).to(self.device), "im_file": tuple(self.files[int(i)] for i in ids),
                   "ori_shape": tuple((p.mosaics[0].shapes[0] if p.mosaics[0].shapes else None) for p in plans),
                   "ratio_pad": tuple((p.mosaics[0].shapes[1] if p.mosaics[0].shapes else None) for p in plans)}

    def __iter__(self):
        from PIL import Image

        from . import _lib as L

        lib = L.load()
        order = self._order()
        self.epoch += 1
        if self.augment:
            yield from self._iter_augmented(order)
            return
        S = self.imgsz
        for b0 in range(0, len(order), self.bs):
            ids = order[b0:b0 + self.bs]
            FH, FW = self.frames[b0 // self.bs] if self.frames is not None else (S, S)
            items = (L.LetterboxItem * len(ids))()
            keep, cls, prob, box, bidx, shapes, files = [], [], [], [], [], [], []
            for j, i in enumerate(ids):
                im = np.asarray(Image.open(self.files[i]).convert("RGB"))[:, :, ::-1]  # the kernel takes cv2's BGR order
                lb = self.labels[i]
                xywh, shp, (new_w, new_h, top, left) = letterbox_labels(lb[:, 2:], im.shape[:2], S, (FH, FW))
                t = torch.from_numpy(np.ascontiguousarray(im)).to(self.device, non_blocking=True)
                keep.append(t)
                it = items[j]
                it.img, it.h, it.w, it.pitch = t.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3
                it.new_w, it.new_h, it.top, it.left = new_w, new_h, top, left
                it.area = 1  # load_image of a non-augmented loader: INTER_AREA when shrinking (the kernel keeps bilinear when enlarging)
                cls.append(lb[:, 0:1]), prob.append(lb[:, 1:2]), box.append(xywh), bidx.append(np.full(len(lb), j, np.float32))
                shapes.append(shp), files.append(self.files[i])
            tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
            img = torch.empty((len(ids), 3, FH, FW), dtype=torch.uint8, device=self.device)
            st = torch.cuda.current_stream(self.device)
            L.check(lib.cdet_letterbox_batch(tab.data_ptr(), len(ids), img.data_ptr(), FH, FW, L.U8, 114, st.cuda_stream), "cdet_letterbox_batch")
            for t in keep + [tab]:
                t.record_stream(st)
            cat = lambda xs, w: torch.from_numpy(np.concatenate(xs, 0).reshape(-1, w).astype(np.float32)).to(self.device)  # noqa: E731
            yield {"img": img, "cls": cat(cls, 1), "prob": cat(prob, 1), "bboxes": cat(box, 4), "batch_idx": cat(bidx, 1).reshape(-1),
                   "im_file": tuple(files), "ori_shape": tuple(s[0] for s in shapes), "ratio_pad": tuple(s[1] for s in shapes)}


def datasets_from_yaml(path: str, tasks: Sequence[str], nc: Sequence[int], bs: Sequence[int], imgsz: int, device="cuda", rank: int = 0,
                       world_size: int = 1, augment: bool = False, hyp: Optional[dict] = None, labels_from_xml: bool = False,
                       use_multi_labels: bool = False, use_soft_labels: bool = False, single_cls: bool = False):
    """-> (train {task: TaskDataset}, val {task: TaskDataset} (rank 0 validates, unsharded), names {task: [str]})."""
    import yaml

    d = yaml.safe_load(open(path))
    ids = list(d.get("task_ids", tasks))
    assert list(tasks) == ids, f"--tasks {list(tasks)} does not match the dataset's task_ids {ids}"
    assert [int(v) for v in d["nc"]] == [int(v) for v in nc], f"--nc {list(nc)} does not match the dataset's nc {d['nc']}"
    root = Path(path).resolve().parent

    def res(p):
        return str(p if os.path.isabs(p) else root / p)

    names = {t: [str(n) for n in d["names"][i]] for i, t in enumerate(ids)} if d.get("names") else None
    lab = dict(labels_from_xml=labels_from_xml, use_multi_labels=use_multi_labels, use_soft_labels=use_soft_labels, single_cls=single_cls)
    cn = (lambda t: names[t] if names else None)  # noqa: E731  (the label files' own class names, also with --single-cls)
    train = {t: TaskDataset(res(d["train"][i]), imgsz, bs[i], nc[i], device, rank, world_size, shuffle=True, seed=i, augment=augment, hyp=hyp,
                            classnames=cn(t), balanced=True, **lab) for i, t in enumerate(ids)}  # (train.py:95 balanced_sampler=True)
    # validation loaders of the reference: rectangular batches with pad 0.5, the largest task batch size (utils/train_utils.py:45-57)
    val = ({t: TaskDataset(res(d["val"][i]), imgsz, max(bs), nc[i], device, 0, 1, shuffle=False, classnames=cn(t), rect=True, **lab)
            for i, t in enumerate(ids)} if d.get("val") else None)
    if single_cls and names:  # utils/models_manager.py:87
        names = {t: (["item"] if len(v) != 1 else v) for t, v in names.items()}
    return train, val, names
