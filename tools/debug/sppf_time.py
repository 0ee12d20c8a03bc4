"""time of cdet_sppf_pool on the 20 x 20 x 320-channel SPPF map of the batch-32 forward (three chained 5x5 pools, one launch)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from cerberusdet_amd import ops
for N in (32, 128):
    buf = ops.View(torch.randn(N, 20, 20, 1280, device="cuda").to(torch.bfloat16))
    for _ in range(5):
        ops.sppf_pool(buf, 320)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        ops.sppf_pool(buf, 320)
    e1.record()
    torch.cuda.synchronize()
    print(f"N {N}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per launch")
