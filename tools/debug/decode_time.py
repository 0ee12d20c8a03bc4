"""time of cdet_detect_decode on the batch-32 head maps (8400 anchors, nc 20)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from cerberusdet_amd import ops
for N in (32, 128):
    for ld in (88, 85):
        feats = [torch.randn(N, h, h, ld, device="cuda") for h in (80, 40, 20)]
        for _ in range(5):
            ops.detect_decode(feats, 20, (8.0, 16.0, 32.0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            ops.detect_decode(feats, 20, (8.0, 16.0, 32.0))
        e1.record()
        torch.cuda.synchronize()
        print(f"N {N} ld {ld} ({'four lanes per anchor' if ld % 4 == 0 else 'one thread per anchor'}): {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per call (incl. the output allocation)")
