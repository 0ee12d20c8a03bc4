"""North-star forward with / without the fused first two backbone rows (CDET_STEM_FUSE), and the fused launch alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
half = os.environ.get("HALF", "0") == "1"
bs = int(os.environ.get("BS", "32"))
x = torch.rand(bs, 3, 640, 640)
x = (x.half() if half else x.bfloat16()).to(dev)
model.eval()
model = model.half() if half else model.bfloat16()


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
for fuse in ("1", "0", "1", "0"):
    os.environ["CDET_STEM_FUSE"] = fuse
    model._plans = {}
    with torch.no_grad():
        for _ in range(30):
            model(x)
        res[fuse] = [t.float().clone() for t in model(x)[bench.TASKS[0]][1]]
    torch.cuda.synchronize()
    plan = model.get_plan(bench.TASKS, x.shape, x.dtype, training=False)
    fn, args = plan.fwd[0]
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(*args, st)
    e1.record()
    torch.cuda.synchronize()
    print(f"fuse {fuse}: forward {timed()} ms; first launch {fn.__name__} {e0.elapsed_time(e1) / 20:.4f} ms; launches {plan.n_fwd_calls}", flush=True)
print("head maps equal (fused vs two-kernel):", all(torch.equal(a, b) for a, b in zip(res["1"], res["0"])))
