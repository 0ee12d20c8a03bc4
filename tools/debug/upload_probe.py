"""Which host call of CerberusPreprocessor.preprocess waits for the previous batch's GPU work (predict_stream)?"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402


def main():
    from cerberusdet_amd.cerberusdet_inference import CerberusDetInference

    device = torch.device("cuda", 0)
    model, cfg = bench.build_model("v8x_2task.yaml", device)
    det = CerberusDetInference(model, device=str(device), half=True, img_size=640)
    x = torch.zeros(32, 3, 640, 640, device=device, dtype=torch.float16)
    frame = np.zeros((720, 1280, 3), np.uint8)
    pinned = torch.from_numpy(frame).pin_memory()
    small = bytearray(2048)
    for kind in ("plain", "priority"):
        side = torch.cuda.Stream(device) if kind == "plain" else torch.cuda.Stream(device, priority=-1)
        with torch.no_grad():
            for rep in range(3):
                det.predict(x)
                torch.cuda.synchronize()
                p = det.predict_async(x)  # ~14 ms of GPU work now pending on the current stream
                t = [time.perf_counter()]
                with torch.cuda.stream(side):
                    a = pinned.to(device, non_blocking=True); t.append(time.perf_counter())
                    b = torch.from_numpy(frame).to(device, non_blocking=True); t.append(time.perf_counter())
                    c = torch.frombuffer(small, dtype=torch.uint8).to(device); t.append(time.perf_counter())
                    d = torch.empty((32, 3, 640, 640), dtype=torch.float16, device=device); t.append(time.perf_counter())
                    d.zero_(); t.append(time.perf_counter())
                side.synchronize(); t.append(time.perf_counter())
                p.result(); t.append(time.perf_counter())
                names = ["pinned H2D", "pageable 2.7MB H2D", "pageable 2KB H2D", "empty", "kernel launch", "side.synchronize", "result()"]
                print(kind, rep, {n: round((t[i + 1] - t[i]) * 1e3, 2) for i, n in enumerate(names)})
                del a, b, c, d


if __name__ == "__main__":
    main()
