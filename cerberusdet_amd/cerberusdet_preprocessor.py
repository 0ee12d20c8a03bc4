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
        self._side = {}   # device -> upload / letterbox stream
        self._stage = {}  # device -> ring of pinned staging buffers
        self._threads = None

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
        for im in images:
            assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3, "images must be uint8 HWC BGR"
        cur = torch.cuda.current_stream(device)
        out_dtype = torch.float16 if self.half else torch.float32
        if cur.query():
            # Nothing pending on the caller's stream (the synchronous predict loop): per-frame copies straight from the caller's arrays --
            # the runtime's pageable path moves 32 720p frames in ~2 ms on an idle GPU.
            keep = []
            for it, im, g in zip(items, images, geo):
                t = torch.from_numpy(np.ascontiguousarray(im)).to(device, non_blocking=True)
                keep.append(t)
                it.img, it.h, it.w, it.pitch = t.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3
                it.new_w, it.new_h, it.top, it.left = g[0], g[1], g[2], g[4]
            tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
            out = torch.empty((len(images), 3, H, W), dtype=out_dtype, device=device)
            L.check(lib.cdet_letterbox_batch(tab.data_ptr(), len(images), out.data_ptr(), H, W, L.F16 if self.half else L.F32, 114, cur.cuda_stream),
                    "cdet_letterbox_batch")
            for t in keep + [tab]:  # the caching allocator must not hand these buffers out before the kernel has read them
                t.record_stream(cur)
            return out
        # The previous batch is still running (CerberusDetInference.predict_stream keeps batches in flight). Copies from pageable memory
        # are host-synchronous and each of them then queues for the busy GPU: 14 ms instead of 2 for 32 frames (tools/debug/upload_probe2.py).
        # So: ONE upload per batch -- frames and descriptor table packed into a pinned staging buffer (host memcpy on a few threads), a single
        # asynchronous copy and the letterbox kernel on a side stream of the pre-processor; the caller's stream waits for that kernel.
        offs, need = [], 0
        for im in images:
            offs.append(need)
            need += (im.nbytes + 255) // 256 * 256
        tab_off, tab_bytes = need, C.sizeof(items)
        need += (tab_bytes + 255) // 256 * 256
        side = self._side.get(device)
        if side is None:
            side = self._side[device] = torch.cuda.Stream(device, priority=-1)
        stage, ev = self._staging(device, need)
        ev.synchronize()  # the copy that last read this staging buffer has finished (three buffers rotate)
        host = stage.numpy()
        with torch.cuda.stream(side):
            dev = torch.empty(need, dtype=torch.uint8, device=device)
            base = dev.data_ptr()
            list(self._pool().map(lambda a: np.copyto(host[a[1]:a[1] + a[0].nbytes].reshape(a[0].shape), a[0]), zip(images, offs)))
            for it, im, g, o in zip(items, images, geo, offs):
                it.img, it.h, it.w, it.pitch = base + o, im.shape[0], im.shape[1], im.shape[1] * 3
                it.new_w, it.new_h, it.top, it.left = g[0], g[1], g[2], g[4]
            C.memmove(stage.data_ptr() + tab_off, C.addressof(items), tab_bytes)
            dev.copy_(stage[:need], non_blocking=True)
            ev.record(side)
            out = torch.empty((len(images), 3, H, W), dtype=out_dtype, device=device)
            L.check(lib.cdet_letterbox_batch(base + tab_off, len(images), out.data_ptr(), H, W, L.F16 if self.half else L.F32, 114, side.cuda_stream),
                    "cdet_letterbox_batch")
        cur.wait_stream(side)
        out.record_stream(cur)  # allocated on the side stream, consumed on the caller's
        return out

    def _pool(self):
        if self._threads is None:
            from concurrent.futures import ThreadPoolExecutor

            self._threads = ThreadPoolExecutor(max_workers=4, thread_name_prefix="cdet-stage")  # numpy's copy releases the GIL
        return self._threads

    def _staging(self, device, need):
        """Next of three rotating pinned staging buffers (grown on demand) and the event of the copy that last read it."""
        ring = self._stage.setdefault(device, {"i": 0, "bufs": [None] * 3})
        ring["i"] = (ring["i"] + 1) % 3
        slot = ring["bufs"][ring["i"]]
        if slot is None or slot[0].numel() < need:
            if slot is not None:
                slot[1].synchronize()
            slot = ring["bufs"][ring["i"]] = (torch.empty(max(need, 1 << 20), dtype=torch.uint8).pin_memory(), torch.cuda.Event())
        return slot
