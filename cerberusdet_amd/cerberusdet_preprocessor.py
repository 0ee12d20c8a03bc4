"""`CerberusPreprocessor` with the reference's interface (ai-forever/CerberusDet cerberusdet/cerberusdet_preprocessor.py:12-74):
list of BGR uint8 HWC images -> [B, 3, H, W] RGB tensor in [0, 1] on the device. The reference letterboxes every image with
cv2 on the host, transposes, stacks, uploads and divides; here the raw uint8 images are uploaded and ONE HIP kernel
(csrc/preprocess.hip) does resize + border + channel flip + layout + scaling for the whole batch."""
from __future__ import annotations

import ctypes as C
import math
from typing import List

import numpy as np
import torch

from . import _lib as L


def check_img_size(img_size: int, s: int = 32) -> int:
    new = int(math.ceil(img_size / int(s)) * int(s))
    if new != img_size:
        print(f"WARNING: --img-size {img_size} must be multiple of max stride {s}, updating to {new}")
    return new


def letterbox_geometry(shape_hw, new_shape, auto: bool, stride: int):
    """Integer geometry of letterbox() (reference data/augmentations.py:59-86, scaleup=True, scaleFill=False):
    returns (new_w, new_h, top, bottom, left, right)."""
    r = min(new_shape[0] / shape_hw[0], new_shape[1] / shape_hw[1])
    new_w, new_h = int(round(shape_hw[1] * r)), int(round(shape_hw[0] * r))
    dw, dh = new_shape[1] - new_w, new_shape[0] - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    return new_w, new_h, int(round(dh - 0.1)), int(round(dh + 0.1)), int(round(dw - 0.1)), int(round(dw + 0.1))


class CerberusPreprocessor:
    def __init__(self, img_size: int = 640, stride: int = 32, half: bool = False, auto: bool = False):
        self.stride, self.half, self.auto = stride, half, auto
        self.img_size = check_img_size(img_size, s=self.stride)

    def preprocess(self, images: List[np.ndarray], device: torch.device) -> torch.Tensor:
        lib = L.load()
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("cerberusdet_amd pre-processing runs on the MI355X (there is no CPU path)")
        geo = [letterbox_geometry(im.shape[:2], (self.img_size, self.img_size), self.auto, self.stride) for im in images]
        sizes = {(g[1] + g[2] + g[3], g[0] + g[4] + g[5]) for g in geo}
        if len(sizes) != 1:  # the reference's np.stack raises here as well (auto=True with differently shaped images)
            raise ValueError(f"all input arrays must have the same shape after letterbox, got {sorted(sizes)}")
        (H, W), = sizes
        items = (L.LetterboxItem * len(images))()
        keep = []
        for it, im, g in zip(items, images, geo):
            assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3, "images must be uint8 HWC BGR"
            t = torch.from_numpy(np.ascontiguousarray(im)).to(device, non_blocking=True)
            keep.append(t)
            it.img, it.h, it.w, it.pitch = t.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3
            it.new_w, it.new_h, it.top, it.left = g[0], g[1], g[2], g[4]
        tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
        out = torch.empty((len(images), 3, H, W), dtype=torch.float16 if self.half else torch.float32, device=device)
        L.check(lib.cdet_letterbox_batch(tab.data_ptr(), len(images), out.data_ptr(), H, W, L.F16 if self.half else L.F32, 114,
                                         torch.cuda.current_stream(device).cuda_stream), "cdet_letterbox_batch")
        for t in keep:  # the caching allocator must not hand these buffers out before the kernel has read them
            t.record_stream(torch.cuda.current_stream(device))
        tab.record_stream(torch.cuda.current_stream(device))
        return out
