"""North-star forward with the head chains started early (CDET_EARLY_HEADS) or at the head block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
x = torch.rand(32, 3, 640, 640).bfloat16().to(dev)
model.eval().bfloat16()


def timed(reps=30, n=3):
    out = []
    with torch.no_grad():
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                model(x)
            e1.record()
            torch.cuda.synchronize()
            out.append(round(e0.elapsed_time(e1) / reps, 3))
    return out


res = {}
for early in ("1", "0", "1", "0"):
    os.environ["CDET_EARLY_HEADS"] = early
    model._plans = {}
    with torch.no_grad():
        for _ in range(30):
            model(x)
        o = model(x)
        res[early] = [t.float().clone() for tk in bench.TASKS for t in o[tk][1]] + [o[tk][0].float().clone() for tk in bench.TASKS]
    torch.cuda.synchronize()
    print(f"early heads {early}: forward {timed()} ms", flush=True)
print("outputs equal:", all(torch.equal(a, b) for a, b in zip(res["1"], res["0"])))
