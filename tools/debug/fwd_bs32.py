"""North-star check: YOLOv8x 2-task all-heads FORWARD at batch 32 @640 (eval, BN folded, fp16 and bf16 storage): ms and MFMA fraction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
for name, half in (("fp16", True), ("bf16", False)):
    m = model.eval()
    m = m.half() if half else m.bfloat16()
    x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(3))
    x = (x.half() if half else x.bfloat16()).to(dev)
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    tf = 381.31e9 * 32 / dt / 1e12
    print(f"{name}: {dt * 1e3:.2f} ms per batch of 32 (2 heads), {32 / dt:.0f} images/s, {tf:.0f} TF/s = {tf / 2500 * 100:.1f} % of the MFMA peak")
