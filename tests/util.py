"""Shared test helpers: golden loading, oracle model construction from a fixture's meta."""
import json
from pathlib import Path

import numpy as np
import torch

import synth  # tests/golden/synth.py
from oracle import graph as og

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
# worst tensor only has a loose sanity bound.
WC_BOUNDS = dict(map_rel_l2=0.012, items_rtol=0.02, grad_rel_l2_median=0.08, grad_rel_l2_p98=0.4, grad_rel_l2_worst=0.8,
                 grad_cos_p02=0.93, grad_cos_p10=0.97, grad_cos_worst=0.7)


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
