"""YOLOv8x 3-task (VOC + O365 animals + O365 tableware) training iterations at batch 32 per task @640: sanity + images/s."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, yaml
import bench
from cerberusdet_amd.models import CerberusDet
from cerberusdet_amd.trainers import Averaging

dev = torch.device("cuda", 0)
tasks, ncs = ["voc", "objects365_animals", "objects365_tableware"], [20, 19, 12]
cfg = yaml.safe_load(open(bench.ROOT / "cerberusdet_amd" / "models" / "cfg" / "v8x_3task.yaml"))
torch.manual_seed(0)
m = CerberusDet(tasks, ncs, cfg=cfg, verbose=False)
m.sequential_split(cfg["cerber"], "cpu")
m.hyp = bench.HYP
m = m.to(dev).train()
hyp = dict(bench.HYP)
for k in ("box", "cls", "dfl"):
    if isinstance(hyp[k], (list, tuple)):
        hyp[k] = list(hyp[k]) + [hyp[k][-1]] * (3 - len(hyp[k]))
tr = Averaging(dev, m, hyp, tasks)
data = {t: bench.synth_batch(0, i, 0, 32, ncs[i], 640, dev) for i, t in enumerate(tasks)}
for _ in range(3):
    out = tr.train_step(data, n_max=8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = tr.train_step(data, n_max=8)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print({t: [round(float(v), 3) for v in out[t][:4]] for t in tasks})
print(f"3-task: {dt * 1e3:.1f} ms per iteration, {96 / dt:.1f} images/s; plans:", {t: m.execution_plan([t])[0][:4] for t in tasks})
