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
