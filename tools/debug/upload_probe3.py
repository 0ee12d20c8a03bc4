"""Upload variants while a forward is pending: 32 pageable copies, 32 pinned copies, one pinned copy of the stacked batch (+ the host stack)."""
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
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8) for _ in range(32)]
    pinned = [torch.from_numpy(f).pin_memory() for f in frames]
    big = torch.empty((32, 720, 1280, 3), dtype=torch.uint8).pin_memory()
    side = torch.cuda.Stream(device, priority=-1)
    dst = torch.empty((32, 720, 1280, 3), dtype=torch.uint8, device=device)

    def T():
        return time.perf_counter()

    with torch.no_grad():
        for busy in (False, True, True):
            det.predict(x)
            torch.cuda.synchronize()
            for name in ("32 pageable", "32 pinned", "host stack + 1 pinned", "host loop copy + 1 pinned"):
                p = det.predict_async(x) if busy else None
                t0 = T()
                with torch.cuda.stream(side):
                    if name == "32 pageable":
                        k = [torch.from_numpy(f).to(device, non_blocking=True) for f in frames]
                    elif name == "32 pinned":
                        k = [f.to(device, non_blocking=True) for f in pinned]
                    elif name == "host stack + 1 pinned":
                        torch.stack([torch.from_numpy(f) for f in frames], out=big)
                        t_h = T()
                        dst.copy_(big, non_blocking=True)
                    else:
                        for i, f in enumerate(frames):
                            big[i].copy_(torch.from_numpy(f))
                        t_h = T()
                        dst.copy_(big, non_blocking=True)
                t1 = T()
                side.synchronize()
                t2 = T()
                if p is not None:
                    p.result()
                t3 = T()
                extra = f" (host part {1e3 * (t_h - t0):.2f})" if "host" in name else ""
                print(f"busy={busy} {name:28s} enqueue {1e3 * (t1 - t0):6.2f} ms{extra}, until done {1e3 * (t2 - t0):6.2f} ms, forward done after {1e3 * (t3 - t0):6.2f}")
                torch.cuda.synchronize()


if __name__ == "__main__":
    main()
