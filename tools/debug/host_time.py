"""Host enqueue time of one training iteration (python + ctypes launches) vs the GPU time it has to cover."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cerberusdet_amd.trainers import Averaging

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
tr = Averaging(dev, model, bench.HYP, bench.TASKS)
data = [{t: bench.synth_batch(0, ti, i, int(os.environ.get("BS", "32")), [20, 19][ti], 640, dev) for ti, t in enumerate(bench.TASKS)} for i in range(2)]
for i in range(4):
    tr.train_step(data[i % 2], n_max=8)
torch.cuda.synchronize()
host, total = [], []
for i in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(data[i % 2], n_max=8)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("host enqueue ms/step", sum(host) / len(host), " wall ms/step", sum(total) / len(total))
