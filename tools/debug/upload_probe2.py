"""Phase timers inside a copy of CerberusPreprocessor.preprocess while the previous batch's forward is pending."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402


def main():
    from cerberusdet_amd import _lib as L
    from cerberusdet_amd.cerberusdet_inference import CerberusDetInference
    from cerberusdet_amd.cerberusdet_preprocessor import letterbox_geometry

    lib = L.load()
    device = torch.device("cuda", 0)
    model, cfg = bench.build_model("v8x_2task.yaml", device)
    det = CerberusDetInference(model, device=str(device), half=True, img_size=640)
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8) for _ in range(32)]
    side = torch.cuda.Stream(device, priority=-1)

    def pre(images, T):
        geo = [letterbox_geometry(im.shape[:2], (640, 640), False, 32) for im in images]
        H = W = 640
        items = (L.LetterboxItem * len(images))()
        keep = []
        cur = torch.cuda.current_stream(device)
        T.append(("geo", time.perf_counter()))
        with torch.cuda.stream(side):
            for it, im, g in zip(items, images, geo):
                t = torch.from_numpy(np.ascontiguousarray(im)).to(device, non_blocking=True)
                keep.append(t)
                it.img, it.h, it.w, it.pitch = t.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3
                it.new_w, it.new_h, it.top, it.left = g[0], g[1], g[2], g[4]
            T.append(("32 copies", time.perf_counter()))
            tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
            T.append(("tab", time.perf_counter()))
            out = torch.empty((len(images), 3, H, W), dtype=torch.float16, device=device)
            T.append(("empty", time.perf_counter()))
            L.check(lib.cdet_letterbox_batch(tab.data_ptr(), len(images), out.data_ptr(), H, W, L.F16, 114, side.cuda_stream), "lb")
            T.append(("launch", time.perf_counter()))
        cur.wait_stream(side)
        out.record_stream(cur)
        T.append(("wait_stream", time.perf_counter()))
        return out

    with torch.no_grad():
        x = pre(frames, [])
        det.predict(x)
        torch.cuda.synchronize()
        pend = []
        for k in range(5):
            T = [("start", time.perf_counter())]
            x = pre(frames, T)
            pend.append(det.predict_async(x, original_shape=(720, 1280)))
            T.append(("predict_async", time.perf_counter()))
            if len(pend) >= 2:
                pend.pop(0).result()
            T.append(("result", time.perf_counter()))
            print(k, {n: round((T[i][1] - T[i - 1][1]) * 1e3, 2) for i, (n, _) in enumerate(T) if i})


if __name__ == "__main__":
    main()
