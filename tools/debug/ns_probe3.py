"""north-star forward after training in the same process: does an idle period / the trainer's state change the steady-state time?"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cerberusdet_amd.trainers import Averaging

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
x = torch.rand(32, 3, 640, 640).bfloat16().to(dev)


def blocks(tag, n=8, reps=20):
    model.eval().bfloat16()
    out = []
    with torch.no_grad():
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                model(x)
            e1.record()
            torch.cuda.synchronize()
            out.append(round(e0.elapsed_time(e1) / reps, 2))
    print(tag, out, "mem GB", round(torch.cuda.memory_allocated() / 2**30, 1), round(torch.cuda.memory_reserved() / 2**30, 1), flush=True)
    model.train()


order = os.environ.get("ORDER", "eval_first")
if order == "eval_first":
    blocks("fresh")
model.train().bfloat16()
tr = Averaging(dev, model, bench.HYP, bench.TASKS, epochs=100, nb=1000)
data = {t: bench.synth_batch(0, ti, 0, 32, bench.NC[ti], 640, dev) for ti, t in enumerate(bench.TASKS)}
for _ in range(6):
    tr.train_step(data, n_max=8)
torch.cuda.synchronize()
blocks("after 6 train steps")
time.sleep(5)
blocks("after 5 s idle")
for _ in range(3):
    tr.train_step(data, n_max=8)
torch.cuda.synchronize()
blocks("after 3 more train steps")
time.sleep(1)
blocks("after 1 s idle")
del tr, data
gc.collect()
torch.cuda.empty_cache()
time.sleep(3)
blocks("trainer freed + 3 s idle")
