"""predict_stream on the bench's calibrated detector: inter-result intervals."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402


def main():
    import gc
    if len(sys.argv) > 1 and sys.argv[1] == "nogc":
        gc.disable()
    device = torch.device("cuda", 0)
    model, cfg = bench.build_model("v8x_2task.yaml", device)
    model.eval()
    real = bench.time.perf_counter
    import cerberusdet_amd.cerberusdet_inference as CI

    # run bench.predict_e2e but with a patched predict_stream that logs intervals
    orig = CI.CerberusDetInference.predict_stream

    def logged(self, batches, depth=2, **kw):
        ts = [real()]
        for r in orig(self, batches, depth=depth, **kw):
            ts.append(real())
            yield r
        iv = np.diff(ts) * 1e3
        print("intervals ms:", np.round(iv, 1).tolist())

    CI.CerberusDetInference.predict_stream = logged
    out = bench.predict_e2e(model, device)  # (in-process: the probe wants the patched predict_stream)
    print({k: v for k, v in out.items() if "images" in k}, out["infer_e2e"]["pipelined_ms_per_batch"])


if __name__ == "__main__":
    main()
