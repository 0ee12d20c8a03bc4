"""One-off (round 5): the bench's north-star forward measured 14.7 ms where the stand-alone tools measure 12.8 on the same box.
Which part of the bench's context costs it? Variants: plain model; lanes 1..4 created up front; after a trainer's iterations."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from cerberusdet_amd import engine

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "plain"
model, _ = bench.build_model("v8x_2task.yaml", dev)
if which == "lanes4":
    engine.lane_stream(dev, 4)
if which.startswith("use"):  # use lanes in a given order before the forward ever runs: "use124" = lanes 1, 2, 4
    raw = {k: torch.cuda.Stream() for k in "1234"} if which.endswith("raw") else None  # "...raw": bare streams, bound to queues by first use
    for ch in which[3:].replace("raw", ""):
        with torch.cuda.stream(raw[ch] if raw else engine.lane_stream(dev, int(ch))):
            torch.zeros(1024, device=dev).add_(1)
    torch.cuda.synchronize()
if which in ("trainer", "trainer_nofold"):
    from cerberusdet_amd.trainers import Averaging
    tr = Averaging(dev, model, bench.HYP, bench.TASKS, epochs=100, nb=1000)
    data = {t: bench.synth_batch(0, ti, 0, 32, bench.NC[ti], 640, dev) for ti, t in enumerate(bench.TASKS)}
    for _ in range(4):
        tr.train_step(data, n_max=8)
    torch.cuda.synchronize()
out = bench.north_star_forward(model, dev)
print(which, out["ms"], out["ms_zero_copy"], "streams:", sorted(k[1] for k in engine._LANE_STREAMS))
