"""Where the time of CerberusDetInference.predict_stream goes: host timestamps around preprocess / predict_async / result per batch."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402


def main():
    from cerberusdet_amd.cerberusdet_inference import CerberusDetInference
    from cerberusdet_amd.cerberusdet_preprocessor import CerberusPreprocessor

    device = torch.device("cuda", 0)
    model, cfg = bench.build_model("v8x_2task.yaml", device)
    det = CerberusDetInference(model, device=str(device), half=True, img_size=640, conf_thres=float(sys.argv[1]) if len(sys.argv) > 1 else 0.02)
    for t in det.model.heads:  # plenty of candidates so that NMS / merge / dicts do real work
        for lvl in range(3):
            det.model.get_head(t).cv3[lvl][2].bias.data += 6.0
    det.model.mark_weights_changed()
    pre = CerberusPreprocessor(img_size=640, stride=det.stride, half=True, auto=False)
    rng = np.random.default_rng(11)
    bs = 32
    frames = [rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8) for _ in range(bs)]
    pinned = [torch.from_numpy(f).pin_memory().numpy() for f in frames]
    with torch.no_grad():
        for mode, src in (("pageable", frames), ("pinned", pinned)):
            for _ in range(3):
                det.predict(pre.preprocess(src, device), original_shape=(720, 1280))
            torch.cuda.synchronize()
            rows = []
            pend = []
            t00 = time.perf_counter()
            for k in range(10):
                t0 = time.perf_counter()
                x = pre.preprocess(src, device)
                t1 = time.perf_counter()
                pend.append(det.predict_async(x, original_shape=(720, 1280)))
                t2 = time.perf_counter()
                if len(pend) >= 2:
                    r = pend.pop(0).result()
                t3 = time.perf_counter()
                rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            nres = 0
            for p in pend:
                nres = sum(len(r) for r in p.result()) / bs
            torch.cuda.synchronize()
            tot = (time.perf_counter() - t00) * 1e3 / 10
            print(mode, "per batch", round(tot, 2), "ms; results per image", nres, "; [preprocess, enqueue, result] per iteration:")
            for r in rows:
                print("   ", [round(v, 2) for v in r])
            # sync reference
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(5):
                det.predict(pre.preprocess(src, device), original_shape=(720, 1280))
            torch.cuda.synchronize()
            print(mode, "synchronous per batch", round((time.perf_counter() - t0) * 1e3 / 5, 2))


if __name__ == "__main__":
    main()
