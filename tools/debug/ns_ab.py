import sys, os, torch, statistics
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
x = torch.rand(32, 3, 640, 640).bfloat16().to(dev)
model.eval().bfloat16()
ts = []
with torch.no_grad():
    for _ in range(60): model(x)
    torch.cuda.synchronize()
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): model(x)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30)
print("north-star ms", round(statistics.median(ts), 3), [round(t, 3) for t in ts])
