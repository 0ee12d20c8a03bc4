#!/usr/bin/env python3
"""Golden for the validation arithmetic: runs the REAL reference's val.process_batch and utils.metrics.ap_per_class (build container
only) on seed-derived synthetic predictions / labels (tests/golden/synth.py::val_case) and stores the outputs.

    python tools/make_golden_val.py        # writes tests/golden/val.npz
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402

sys.path.insert(0, str(mg.OUT))
import synth  # noqa: E402


def main():
    if not mg.REF.exists():
        sys.exit("needs /root/reference (build container only)")
    mg._install_stubs()
    sys.path.insert(0, str(mg.REF))
    from cerberusdet.utils.metrics import ap_per_class
    from cerberusdet.val import process_batch

    out = {}
    iouv = torch.linspace(0.5, 0.95, 10)
    out["iouv"] = iouv.numpy()
    stats = []
    for ci, (seed, n, m, nc) in enumerate(synth.VAL_CASES):
        det, lab = synth.val_case(seed, n, m, nc)
        correct = process_batch(torch.from_numpy(det), torch.from_numpy(lab), iouv) if n and m else torch.zeros(n, 10, dtype=torch.bool)
        out[f"case{ci}/correct"] = correct.numpy()
        stats.append((correct.numpy(), det[:, 4], det[:, 5], lab[:, 0]))
    tp, conf, pcls, tcls = [np.concatenate(x, 0) for x in zip(*stats)]
    r = ap_per_class(tp, conf, pcls, tcls, plot=False, names={})
    for k, v in zip(("tp", "fp", "p", "r", "f1", "ap", "classes"), r):
        out[f"ap/{k}"] = np.asarray(v)
    np.savez_compressed(mg.OUT / "val.npz", **out)
    print({k: v.shape for k, v in out.items()}, (mg.OUT / "val.npz").stat().st_size / 1024, "KiB")
    print("map50", out["ap/ap"][:, 0].mean(), "map", out["ap/ap"].mean(), "correct@.5 per case", [int(out[f"case{i}/correct"][:, 0].sum()) for i in range(len(synth.VAL_CASES))])


if __name__ == "__main__":
    main()
