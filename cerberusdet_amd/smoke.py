"""smoke(): one tiny pass of the whole hot path on cuda:0, checked against the CPU oracle (tests infrastructure is allowed here):
compiled eval forward + batched NMS, and one fused train iteration (forward, HIP loss, backward, fused optimizer)."""
import copy
import sys
from pathlib import Path

import numpy as np
import torch


def run():
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    import yaml

    from oracle import graph as og
    from oracle import nms as on
    from .models import CerberusDet
    from .trainers import Averaging
    from .utils.general import non_max_suppression

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    dev = torch.device("cuda", 0)
    cfg = yaml.safe_load(open(root / "cerberusdet_amd" / "models" / "cfg" / "v8n_2task.yaml"))
    tasks, nc = ["voc", "objects365_animals"], [20, 19]
    torch.manual_seed(0)
    model = CerberusDet(tasks, nc, cfg=copy.deepcopy(cfg), verbose=False)
    model.sequential_split(cfg["cerber"], "cpu")
    g = og.build_graph(cfg, tasks, nc)
    og.apply_cerber_schedule(g, cfg["cerber"])
    w = og.init_weights(g, seed=3)
    model.load_state_dict(w)
    model.hyp = dict(box=[7.5, 7.5], cls=[0.5, 0.5], dfl=[1.5, 1.5], lr0=0.00309, lrf=0.0956, momentum=0.952, weight_decay=0.00037,
                     warmup_epochs=2.04, warmup_momentum=0.898, warmup_bias_lr=0.0502)
    model = model.to(dev)
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(1))
    # ---- eval forward vs oracle
    model.eval()
    with torch.no_grad():
        out = model(x.to(dev))
        ref = og.forward(g, w, x, None, training=False)
    for t in tasks:
        y, yr = out[t][0].float().cpu().numpy(), ref[t][0].numpy()
        err = np.linalg.norm(y - yr) / np.linalg.norm(yr)
        assert err < 3e-2, f"eval forward mismatch for {t}: rel-L2 {err:.4f}"
    # ---- the same forward at the reference's precision (model.full_precision(), precise.py) vs the fp32 oracle: tight
    model.full_precision()
    with torch.no_grad():
        outp = model(x.to(dev))
    for t in tasks:
        y, yr = outp[t][0].cpu().numpy(), ref[t][0].numpy()
        err = np.abs(y[:, :4] - yr[:, :4]).max() / np.abs(yr[:, :4]).max()
        assert err < 1e-3 and np.abs(y[:, 4:] - yr[:, 4:]).max() < 1e-3, f"full-precision forward mismatch for {t}: {err:.2e}"
    model.bfloat16()
    # ---- batched NMS vs oracle (bit exact)
    import importlib.util

    spec = importlib.util.spec_from_file_location("synth", root / "tests" / "golden" / "synth.py")
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    yp = synth.synth_pred(2, 20, 2100, 100, 5)
    got = non_max_suppression(torch.from_numpy(yp).to(dev), 0.25, 0.45)
    want = on.non_max_suppression(yp, 0.25, 0.45)
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b), "NMS mismatch"
    # ---- one fused training iteration
    model.train()
    tr = Averaging(dev, model, model.hyp, tasks, epochs=3, nb=10)
    batches = {}
    for ti, t in enumerate(tasks):
        b = synth.make_batch(2, 3, nc[ti], 40 + ti)
        batches[t] = dict(img=(x * 255).to(torch.uint8).to(dev), **{k: torch.from_numpy(v).to(dev) for k, v in b.items()})
    w0 = model.blocks[0].model[1].bn.weight.detach().clone()  # BN weights move on step 0 (warm-up quirk: conv lr starts at 0)
    items = tr.train_step(batches)
    torch.cuda.synchronize()
    for t in tasks:
        v = items[t].cpu().numpy()
        assert np.isfinite(v).all() and v[3] > 0, f"bad loss for {t}: {v}"
    assert not torch.equal(w0, model.blocks[0].model[1].bn.weight), "optimizer step did not update the weights"
    print("smoke OK:", {t: [round(float(z), 4) for z in items[t].tolist()] for t in tasks})
