"""Time of the fused optimizer step (grad_sqnorm + sgd_ema) on the YOLOv8x 2-task parameter set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cerberusdet_amd.trainers import Averaging

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
tr = Averaging(dev, model, bench.HYP, bench.TASKS, epochs=100, nb=1000)
data = {t: bench.synth_batch(0, ti, 0, 32, bench.NC[ti], 640, dev) for ti, t in enumerate(bench.TASKS)}
for _ in range(2):
    tr.train_step(data, n_max=8)
lrs, mom = tr.lrs(5000, 0)
for _ in range(3):
    tr.optimizer_step(lrs, mom)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    tr.optimizer_step(lrs, mom)
e1.record()
torch.cuda.synchronize()
n = sum(m["p"].numel() for m in tr.slots_meta)
ms = e0.elapsed_time(e1) / 20
print(f"optimizer step: {ms:.3f} ms for {n / 1e6:.1f} M slot elements")
