"""Per-block forward / backward time of one task pass (sequential schedule, HIP events around each block's launches)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cerberusdet_amd import _lib as L
from cerberusdet_amd.trainers import Averaging

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
tr = Averaging(dev, model, bench.HYP, bench.TASKS, task_streams=False)
t = bench.TASKS[0]
b = bench.synth_batch(0, 0, 0, 32, 20, 640, dev)
for _ in range(3):
    tr.forward_backward(t, b, n_max=8, active_tasks=[t])
torch.cuda.synchronize()
plan = model.get_plan(t, b["img"].shape, b["img"].dtype, training=True)
st = torch.cuda.current_stream().cuda_stream
def run(calls):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for fn, args in calls:
        fn(*args, st)
    e1.record()
    return e0, e1
plan.refresh_weights()
ev = []
pos = 0
for idx, end in plan.fwd_marks:
    ev.append(("fwd", idx, len(plan.fwd[pos:end]), run(plan.fwd[pos:end])))
    pos = end
for idx, calls in plan.bwd_groups:
    ev.append(("bwd", idx, len(calls), run(calls)))
torch.cuda.synchronize()
tot = {}
for kind, idx, n, (e0, e1) in ev:
    ms = e0.elapsed_time(e1)
    tot.setdefault(idx, [0, 0, 0])
    tot[idx][0 if kind == "fwd" else 1] += ms
    tot[idx][2] += n
for idx, (f, bw, n) in tot.items():
    print(f"block {idx:2d} {type(model.blocks[idx]).__name__:10s} fwd {f:6.2f} ms  bwd {bw:6.2f} ms  launches {n}")
print("total fwd", sum(v[0] for v in tot.values()), "bwd", sum(v[1] for v in tot.values()))
