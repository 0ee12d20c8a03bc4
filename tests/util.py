"""Shared test helpers: golden loading, oracle model construction from a fixture's meta."""
import json
from pathlib import Path

import numpy as np
import torch

import synth  # tests/golden/synth.py
from oracle import graph as og
from oracle import loss as ol
from oracle import optim as oo

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_golden(name):
    arrays = dict(np.load(GOLDEN / f"{name}.npz"))
    meta = json.load(open(GOLDEN / f"{name}.json"))
    return arrays, meta


def oracle_model_from_meta(meta):
    """Build the oracle graph for a model fixture and its weights (synth.det_tensor by key)."""
    g = og.build_graph(meta["cfg"], meta["tasks"], meta["nc"])
    og.apply_cerber_schedule(g, meta["cfg"].get("cerber", []))
    shapes = og.param_shapes(g)
    w = {k: torch.from_numpy(synth.det_tensor(meta["seed"], k, s)) for k, s in shapes.items()}
    return g, w


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def oracle_wc_model(meta):
    """Oracle graph + synth.det_tensor_wc weights of the well-conditioned train fixture (tests/golden/train_wc.*)."""
    g = og.build_graph(meta["cfg"], meta["tasks"], meta["nc"])
    og.apply_cerber_schedule(g, meta["cfg"].get("cerber", []))
    w = {k: torch.from_numpy(synth.det_tensor_wc(meta["seed"], k, s)) for k, s in og.param_shapes(g).items()}
    return g, w


def update_error(got, ref, start=None):
    """(relative L2 error of the update got - start against ref - start, cosine of the two updates) of one (sampled) tensor."""
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    start = np.zeros_like(ref) if start is None else np.asarray(start, np.float64).ravel()
    d_ref, d_got = ref - start, got - start
    cos = float(d_ref @ d_got / (np.linalg.norm(d_ref) * np.linalg.norm(d_got) + 1e-30))
    return float(np.linalg.norm(d_got - d_ref) / (np.linalg.norm(d_ref) + 1e-30)), cos


# What 16-bit storage costs on the train_wc fixture, measured by emulating bf16 storage of activations and GEMM operands in the fp32
# CPU oracle (test_train_wc_fixture_oracle_matches_reference_and_is_well_conditioned asserts these hold for the emulation): the bounds
# the HIP path is held to on the same fixture. (model_tiny2's train fixture sits at maps 5-12 %, cosines 0.3-0.8 under the same emulation.)
# The bounds are on QUANTILES over the ~350 gradient tensors: the single worst tensor is a heavy-tailed statistic -- it is always one of
# the 8- or 16-element BatchNorm gradients of the first backbone rows, at the far end of the backward path, and whether it lands at
# 0.46 (emulation), 0.49 or 0.54 (two HIP builds that differ only in fp32 summation order) is the realisation of that noise, so the
# worst tensor only has a loose sanity bound. These fixed numbers are the coarse gate; the tight one is wc_compare(): the HIP path's error
# quantiles against the emulation's own, computed in the same test run.
WC_BOUNDS = dict(map_rel_l2=0.012, items_rtol=0.02, grad_rel_l2_median=0.08, grad_rel_l2_p98=0.5, grad_rel_l2_worst=0.8,
                 grad_cos_p02=0.9, grad_cos_p10=0.95, grad_cos_worst=0.7)


def wc_pass(g, w, meta, ti, seeds, emulate=False):
    """One task pass of the train_wc fixture through the oracle: (maps, items, scalar, grads, bn updates)."""
    t, nc = meta["tasks"][ti], meta["nc"]
    x = torch.from_numpy(synth.det_image(seeds[0] + ti, meta["bs"], meta["imgsz"]))
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(meta["bs"], meta["boxes_per_img"], nc[ti], seeds[1] + ti).items()}
    wt = {k: (v.clone().requires_grad_(True) if oo.is_trainable(k) else v) for k, v in w.items()}
    rnd = None
    wf = wt
    if emulate:  # 16-bit storage of activations and GEMM operands, as the HIP path keeps them
        rnd = lambda y: y.to(torch.bfloat16).float()  # noqa: E731
        wf = {k: (v.to(torch.bfloat16).float() if k.endswith(("conv.weight", ".2.weight")) and "dfl" not in k else v) for k, v in wt.items()}
    upd = {}
    feats = og.forward(g, wf, x, t, training=True, bn_updates=upd, act_round=rnd)
    hyp = meta["hyp"]
    scalar, items = ol.detection_loss(feats, batch, nc[ti], dict(box=hyp["box"][ti], cls=hyp["cls"][ti], dfl=hyp["dfl"][ti]))
    scalar.backward()
    grads = {k: v.grad for k, v in wt.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    return [f.detach() for f in feats], items.detach(), float(scalar), grads, upd



def wc_emulation_errors(arrays, meta):
    """Per-tensor (rel-L2, cosine, name) of the fp32 oracle run WITH bf16 storage emulated against the reference's gradients (part A of
    train_wc): the measured price of 16-bit storage on this fixture, which the HIP path is compared with."""
    g, w = oracle_wc_model(meta)
    errs = []
    for ti, t in enumerate(meta["tasks"]):
        _, _, _, eg, _ = wc_pass(g, w, meta, ti, (300, 400), emulate=True)
        keys = [k[len(f"A/{t}/grad/"):] for k in arrays if k.startswith(f"A/{t}/grad/")]
        errs += [update_error(synth.sample(eg[k].numpy()), arrays[f"A/{t}/grad/{k}"]) + (f"{t}:{k}",) for k in keys]
    return errs


def wc_compare(errs, emu, what="", factor=2.0):
    """The HIP path's per-tensor deviations from the reference vs the emulation's, quantile by quantile: rel-L2 (median, p90, p98) and
    1 - cosine (median, p90, p98) may be at most `factor` x the emulation's (+ a small absolute floor). The two are different
    realisations of the same rounding noise, so single tensors are not compared."""
    r, c = np.array([e[0] for e in errs]), 1.0 - np.array([e[1] for e in errs])
    re_, ce = np.array([e[0] for e in emu]), 1.0 - np.array([e[1] for e in emu])
    line = f"{what}: {len(errs)} tensors;"
    ok = True
    for name, a, b, floor in (("rel-L2", r, re_, 0.01), ("1-cos", c, ce, 1e-3)):
        for q in (0.5, 0.9, 0.98):
            x, y = float(np.quantile(a, q)), float(np.quantile(b, q))
            line += f" {name} p{int(q * 100)} {x:.4f} (emulation {y:.4f})"
            ok = ok and x <= factor * y + floor
    assert ok, line
    return line


def wc_check(errs, what=""):
    """errs: [(rel-L2, cosine, name)] of every gradient / update tensor -> assert WC_BOUNDS, return a one-line summary."""
    r = np.array([e[0] for e in errs])
    c = np.array([e[1] for e in errs])
    worst = max(errs)
    line = (f"{what}: {len(errs)} tensors, rel-L2 median {np.median(r):.3f} p98 {np.quantile(r, 0.98):.3f} worst {r.max():.3f} ({worst[2]}), "
            f"cosine worst {c.min():.4f} p02 {np.quantile(c, 0.02):.4f} p10 {np.quantile(c, 0.1):.4f} median {np.median(c):.5f}")
    B = WC_BOUNDS
    assert np.median(r) <= B["grad_rel_l2_median"] and np.quantile(r, 0.98) <= B["grad_rel_l2_p98"] and r.max() <= B["grad_rel_l2_worst"], line
    assert c.min() >= B["grad_cos_worst"] and np.quantile(c, 0.02) >= B["grad_cos_p02"] and np.quantile(c, 0.1) >= B["grad_cos_p10"], line
    return line
