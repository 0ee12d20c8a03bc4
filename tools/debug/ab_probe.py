"""A/B of one plan-compiler switch on the north-star forward inside ONE process (same box, same clocks): python tools/debug/ab_probe.py CDET_HEAD_MERGE [rounds]
The plan is rebuilt for every arm; arms alternate 1, 0, 1, 0, ...; prints per-arm medians."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

sw = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
model, cfg = bench.build_model("v8x_2task.yaml", dev)
x = torch.rand(32, 3, 640, 640).bfloat16().to(dev)
model.eval().bfloat16()
res = {"1": [], "0": []}
for r in range(rounds):
    for arm in ("1", "0"):
        os.environ[sw] = arm
        model._plans = {}
        with torch.no_grad():
            for _ in range(40):
                model(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                model(x)
            e1.record()
            torch.cuda.synchronize()
        res[arm].append(e0.elapsed_time(e1) / 40)
for arm in ("1", "0"):
    print(f"{sw}={arm}: median {statistics.median(res[arm]):.3f} ms  all {[round(v, 3) for v in res[arm]]}", flush=True)
