"""North-star forward (YOLOv8x 2-task all heads, eval, bf16, batch 32 @640) with the first backbone rows run per slice of the batch
(engine.Plan._sliced_stage): CDET_EVAL_SLICE x CDET_EVAL_SLICE_ROWS sweep in one process, default model(x) and zero-copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
bs = int(os.environ.get("BS", "32"))
half = os.environ.get("HALF", "0") == "1"
x = torch.rand(bs, 3, 640, 640)
x = (x.half() if half else x.bfloat16()).to(dev)
model.eval()
model = model.half() if half else model.bfloat16()


def timed(zero_copy, reps=30, n=3):
    out = []
    with torch.no_grad():
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                model(x, zero_copy=zero_copy)
            e1.record()
            torch.cuda.synchronize()
            out.append(round(e0.elapsed_time(e1) / reps, 3))
    return out


combos = [(0, 3)] + [(p, r) for r in (2, 3, 5) for p in (2, 4, 8, 16)] + [(0, 3)]
if len(sys.argv) > 1:
    combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for per, rows in combos:
    os.environ["CDET_EVAL_SLICE"] = str(per)
    os.environ["CDET_EVAL_SLICE_ROWS"] = str(rows)
    model._plans = {}
    with torch.no_grad():
        for _ in range(30):
            model(x, zero_copy=True)
    torch.cuda.synchronize()
    plan = model.get_plan(bench.TASKS, x.shape, x.dtype, training=False)
    print(f"slice {per:2d} rows {rows}: zero-copy {timed(True)}  default {timed(False)}  launches {plan.n_fwd_calls} {getattr(plan, 'sliced', None)}", flush=True)
