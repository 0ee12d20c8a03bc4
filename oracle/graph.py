"""Oracle: functional CPU restatement of the CerberusDet model graph (TEST INFRASTRUCTURE).

Follows (reference paths relative to /root/reference/cerberusdet):
  * YAML -> layers, depth/width scaling ........ models/yolo.py:234-339
  * Conv / Bottleneck / C2f / SPPF / Concat .... models/common.py:51-68,107-117,174-191,230-245,288-295
  * Detect / DFL / bias_init ................... models/yolo.py:48-110
  * block DAG, execution plan .................. models/cerberus.py:212-319,371-403,804-882
  * `cerber` schedule (neck branching) ......... models/cerberus.py:461-737
  * BN constants eps=1e-3, momentum=0.03 ....... utils/torch_utils.py:179-188
  * conv+BN folding ............................ utils/torch_utils.py:191-217

Unlike the reference (nn.Module tree + Controllers) this is a plain-data graph and a
functional interpreter over a flat ``{state_dict_key: tensor}`` dict that uses the
reference's state-dict key schema, so golden fixtures can carry
weights as plain arrays.
"""
from __future__ import annotations

import copy
import math
from typing import Dict, List, Sequence, Union

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # utils/torch_utils.py:184
BN_MOMENTUM = 0.03  # utils/torch_utils.py:185
REG_MAX = 16  # models/yolo.py:76


def make_divisible(x, divisor):  # utils/general.py:206-208
    return math.ceil(x / divisor) * divisor


# ----------------------------------------------------------------------------------------------
# graph construction
# ----------------------------------------------------------------------------------------------
def _layer_spec(m: str, n: int, f, args, ch: List[int], c_in_prev: int, gd, gw, max_ch, nc_all):
    """One YAML row -> (spec dict, c_out). models/yolo.py:283-339."""
    n_ = max(round(n * gd), 1) if n > 1 else n  # depth gain, yolo.py:291
    flist = [f] if isinstance(f, int) else list(f)

    def chan(x):
        return c_in_prev if x == -1 else ch[x]

    if m in ("Conv", "C2f", "SPPF"):
        c1, c2 = chan(flist[0]), args[0]
        if all(c2 != v for v in nc_all):  # yolo.py:311 (quirk: a width equal to some nc is not scaled)
            c2 = make_divisible(min(c2, max_ch) * gw, 8)
        if m == "Conv":
            k = args[1] if len(args) > 1 else 1
            s = args[2] if len(args) > 2 else 1
            return dict(op="Conv", c1=c1, c2=c2, k=k, s=s), c2
        if m == "C2f":
            shortcut = bool(args[1]) if len(args) > 1 else False
            return dict(op="C2f", c1=c1, c2=c2, n=n_, shortcut=shortcut), c2
        k = args[1] if len(args) > 1 else 5
        return dict(op="SPPF", c1=c1, c2=c2, k=k), c2
    if m == "nn.Upsample":
        assert args[1] == 2 and args[2] == "nearest"
        return dict(op="Upsample", scale=2), chan(flist[0])
    if m == "Concat":
        return dict(op="Concat"), sum(chan(x) for x in flist)
    raise NotImplementedError(m)


def build_graph(cfg: dict, task_ids: Sequence[str], nc: Union[int, Sequence[int]]) -> dict:
    """YAML dict -> plain-data graph *before* the `cerber` schedule is applied.

    Node refs: ("blk", i) = output of block i; ("bb", j) = saved output of backbone layer j.
    models/cerberus.py:143-202 (ctor), 212-254 (neck), 256-319 (heads).
    """
    cfg = copy.deepcopy(cfg)
    nc = [nc] * len(task_ids) if isinstance(nc, int) else list(nc)
    gd, gw = cfg["depth_multiple"], cfg["width_multiple"]
    max_ch = cfg.get("max_channels", 1024)
    ch_in = cfg.get("ch", 3)
    nb = len(cfg["backbone"])

    ch: List[int] = []
    bb_layers = []
    c_prev = ch_in
    for j, (f, n, m, args) in enumerate(cfg["backbone"]):
        spec, c2 = _layer_spec(m, n, f, args, ch, c_prev, gd, gw, max_ch, nc)
        spec["f"] = f
        spec["prefix"] = f"blocks.0.model.{j}"
        bb_layers.append(spec)
        ch.append(c2)
        c_prev = c2

    nodes: List[dict] = [dict(idx=0, op="Backbone", layers=bb_layers, inputs=[], task=None, ch_in=ch_in)]
    layer_ind_map = {}
    prev_blk = 0
    for i, (f, n, m, args) in enumerate(cfg["neck"], start=1):
        ind = nb + i - 1
        spec, c2 = _layer_spec(m, n, f, args, ch, c_prev, gd, gw, max_ch, nc)
        flist = [f] if isinstance(f, int) else list(f)
        assert flist[0] == -1 or len(flist) == 1, "Unsupported config"  # cerberus.py:230
        inputs = []
        for x in flist:
            if x == -1:
                if i == 1:
                    raise ValueError("Input for first cerbernet block must be defined")  # cerberus.py:239
                inputs.append(("blk", prev_blk))
            elif x >= nb:
                inputs.append(("blk", layer_ind_map[x]))
            else:
                inputs.append(("bb", x))
        spec.update(idx=i, inputs=inputs, task=None, prefix=f"blocks.{i}")
        nodes.append(spec)
        layer_ind_map[ind] = i
        ch.append(c2)
        c_prev = c2
        prev_blk = i

    assert len(cfg["head"]) == 1  # cerberus.py:259
    f, n, m, args = cfg["head"][0]
    assert m == "Detect"
    heads = {}
    nc_left = list(nc)
    for task in task_ids:
        idx = len(nodes)
        inputs = [("blk", layer_ind_map[x]) for x in f]
        nodes.append(
            dict(idx=idx, op="Detect", inputs=inputs, task=task, nc=nc_left.pop(0), ch=[ch[x] for x in f],
                 prefix=f"blocks.{idx}")
        )
        heads[task] = idx
    return dict(nodes=nodes, heads=heads, nb=nb, cfg=cfg, stride=None)


def _ancestors(nodes, idx, acc):
    for kind, j in nodes[idx]["inputs"]:
        j = 0 if kind == "bb" else j
        if j not in acc:
            acc.add(j)
            _ancestors(nodes, j, acc)
    return acc


def chain(graph, head_idx) -> List[int]:
    """Execution chain of a head: every ancestor in index order, head last (cerberus.py:61-118)."""
    return sorted(_ancestors(graph["nodes"], head_idx, set())) + [head_idx]


def execution_plan(graph, task_ids) -> List[int]:
    """models/cerberus.py:371-403: concatenate each task's chain from its first unscheduled block."""
    if isinstance(task_ids, str):
        task_ids = [task_ids]
    order: List[int] = []
    for t in task_ids:
        c = chain(graph, graph["heads"][t])
        i = 0
        for i, index in enumerate(c):
            if index not in order:
                break
        order += c[i:]
    return order


def apply_cerber_schedule(graph, schedule) -> dict:
    """`cerber: [[neck_idx, [[heads..],[heads..]]], ...]` -> cloned branches.

    models/cerberus.py:704-737 (sequential_split), 635-702 (split), 461-633 (create_nested_branch).
    The first group keeps the original blocks, every other group gets copies of all blocks that
    come after `neck_idx` on its heads' path; copies are appended after the heads in execution
    order (which fixes the state-dict numbering). Returns {new_idx: source_idx} for weight cloning.
    """
    nodes = graph["nodes"]
    heads = graph["heads"]
    head_task = {v: k for k, v in heads.items()}
    schedule = copy.deepcopy(schedule)
    sched_heads = sorted({h for sc in schedule for g in sc[1] for h in g})
    assert sched_heads == sorted(heads.values()) or not sched_heads, f"Invalid cerberusNet config {schedule}"
    clone_src: Dict[int, int] = {}
    for si, (index, groups) in enumerate(schedule):
        nxt = schedule[si + 1:]
        ids_map: Dict[int, Dict[int, int]] = {}
        for sc in nxt:
            for h in [h for g in sc[1] for h in g]:
                ids_map[h] = {sc[0]: (sc[0] if h in groups[0] else None)}
        for a in range(len(groups)):
            for b in range(a + 1, len(groups)):
                if set(groups[a]) & set(groups[b]):
                    raise ValueError("The branching schemes should be disjoint to each other.")
        for group in groups[1:]:
            if index in head_task:
                raise ValueError("Cannot split 's head.")
            names = [head_task[h] for h in group]
            order = execution_plan(graph, names)
            clones: Dict[int, int] = {}
            for ind in order:
                if ind <= index:  # cerberus.py:526
                    continue
                if ind in group:  # cerberus.py:528
                    break
                new = len(nodes)
                node = copy.deepcopy(nodes[ind])
                node["idx"] = new
                node["prefix"] = f"blocks.{new}"
                node["inputs"] = [(k, clones.get(j, j)) if k == "blk" else (k, j) for k, j in node["inputs"]]
                nodes.append(node)
                clones[ind] = new
                clone_src[new] = ind
            for h in group:
                nodes[h]["inputs"] = [(k, clones.get(j, j)) if k == "blk" else (k, j) for k, j in nodes[h]["inputs"]]
                if h in ids_map:
                    for old in list(ids_map[h].keys()):
                        if old in clones:
                            ids_map[h][old] = clones[old]
        for sc in nxt:
            mapped = [ids_map[h][sc[0]] for g in sc[1] for h in g]
            assert None not in mapped and len(set(mapped)) == 1  # cerberus.py:732-733
            sc[0] = mapped[0]
    return clone_src


def serving_tasks(graph) -> Dict[int, List[str]]:
    """block idx -> tasks whose chain contains it (cerberus.py:449-459)."""
    out = {n["idx"]: [] for n in graph["nodes"]}
    for t, h in graph["heads"].items():
        for i in chain(graph, h):
            out[i].append(t)
    return out


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
def _conv_keys(prefix, c1, c2, k):
    return {
        f"{prefix}.conv.weight": (c2, c1, k, k),
        f"{prefix}.bn.weight": (c2,),
        f"{prefix}.bn.bias": (c2,),
        f"{prefix}.bn.running_mean": (c2,),
        f"{prefix}.bn.running_var": (c2,),
        f"{prefix}.bn.num_batches_tracked": (),
    }


def _module_keys(spec, prefix) -> Dict[str, tuple]:
    op = spec["op"]
    if op == "Conv":
        return _conv_keys(prefix, spec["c1"], spec["c2"], spec["k"])
    if op == "C2f":
        c = int(spec["c2"] * 0.5)
        out = {}
        out.update(_conv_keys(f"{prefix}.cv1", spec["c1"], 2 * c, 1))
        out.update(_conv_keys(f"{prefix}.cv2", (2 + spec["n"]) * c, spec["c2"], 1))
        for i in range(spec["n"]):
            out.update(_conv_keys(f"{prefix}.m.{i}.cv1", c, c, 3))
            out.update(_conv_keys(f"{prefix}.m.{i}.cv2", c, c, 3))
        return out
    if op == "SPPF":
        c_ = spec["c1"] // 2
        out = {}
        out.update(_conv_keys(f"{prefix}.cv1", spec["c1"], c_, 1))
        out.update(_conv_keys(f"{prefix}.cv2", c_ * 4, spec["c2"], 1))
        return out
    if op == "Detect":
        nc, ch = spec["nc"], spec["ch"]
        c2, c3 = max(16, ch[0] // 4, REG_MAX * 4), max(ch[0], nc)  # yolo.py:80
        out = {}
        for lvl, x in enumerate(ch):
            out.update(_conv_keys(f"{prefix}.cv2.{lvl}.0", x, c2, 3))
            out.update(_conv_keys(f"{prefix}.cv2.{lvl}.1", c2, c2, 3))
            out[f"{prefix}.cv2.{lvl}.2.weight"] = (4 * REG_MAX, c2, 1, 1)
            out[f"{prefix}.cv2.{lvl}.2.bias"] = (4 * REG_MAX,)
            out.update(_conv_keys(f"{prefix}.cv3.{lvl}.0", x, c3, 3))
            out.update(_conv_keys(f"{prefix}.cv3.{lvl}.1", c3, c3, 3))
            out[f"{prefix}.cv3.{lvl}.2.weight"] = (nc, c3, 1, 1)
            out[f"{prefix}.cv3.{lvl}.2.bias"] = (nc,)
        out[f"{prefix}.dfl.conv.weight"] = (1, REG_MAX, 1, 1)
        return out
    return {}


def param_shapes(graph) -> Dict[str, tuple]:
    """All state-dict keys with shapes, in the reference's schema."""
    out: Dict[str, tuple] = {}
    for node in graph["nodes"]:
        if node["op"] == "Backbone":
            for spec in node["layers"]:
                out.update(_module_keys(spec, spec["prefix"]))
        else:
            out.update(_module_keys(node, node["prefix"]))
    return out


def init_weights(graph, seed=0, strides=(8.0, 16.0, 32.0), randomize_bn=True) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights (numpy MT19937 stream, stable across machines).

    Conv: U(-b, b), b = sqrt(3 / fan_in) (variance-preserving), BN gamma ~ U(0.7, 1.3),
    beta ~ U(-0.2, 0.2), running stats non-trivial so that eval-mode folding is exercised.
    Detect biases follow models/yolo.py:102-110.
    """
    rng = np.random.RandomState(seed)
    w: Dict[str, torch.Tensor] = {}
    heads = {n["prefix"]: n for n in graph["nodes"] if n["op"] == "Detect"}
    for key, shape in param_shapes(graph).items():
        if key.endswith("num_batches_tracked"):
            w[key] = torch.zeros((), dtype=torch.int64)
        elif key.endswith("dfl.conv.weight"):
            w[key] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
        elif key.endswith("conv.weight") or (key.endswith(".2.weight")):
            fan_in = shape[1] * shape[2] * shape[3]
            b = math.sqrt(3.0 / fan_in)
            w[key] = torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))
        elif key.endswith(".2.bias"):
            prefix = key.split(".cv")[0]
            node = heads[prefix]
            lvl = int(key.split(".")[3])
            if ".cv2." in key:
                w[key] = torch.full(shape, 1.0)
            else:
                w[key] = torch.full(shape, math.log(5 / node["nc"] / (640 / strides[lvl]) ** 2))
        elif key.endswith("bn.weight"):
            v = rng.uniform(0.7, 1.3, size=shape) if randomize_bn else np.ones(shape)
            w[key] = torch.from_numpy(v.astype(np.float32))
        elif key.endswith("bn.bias"):
            v = rng.uniform(-0.2, 0.2, size=shape) if randomize_bn else np.zeros(shape)
            w[key] = torch.from_numpy(v.astype(np.float32))
        elif key.endswith("running_mean"):
            v = rng.normal(0, 0.1, size=shape) if randomize_bn else np.zeros(shape)
            w[key] = torch.from_numpy(v.astype(np.float32))
        elif key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)
            w[key] = torch.from_numpy(v.astype(np.float32))
        else:
            raise KeyError(key)
    return w


def clone_branch_weights(weights, clone_src):
    """Weights of cloned blocks = deep copy of their source blocks (cerberus.py:530)."""
    out = dict(weights)
    for new in sorted(clone_src):
        src = clone_src[new]
        sp = f"blocks.{src}."
        for k in list(out.keys()):
            if k.startswith(sp):
                out[f"blocks.{new}." + k[len(sp):]] = out[k].clone()
    return out


def fold_bn(weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Fold every Conv's BN (running stats) into conv weight+bias. utils/torch_utils.py:191-217."""
    out = {}
    for k, v in weights.items():
        if k.endswith(".conv.weight") and k[: -len("conv.weight")] + "bn.weight" in weights:
            p = k[: -len("conv.weight")]
            g, b = weights[p + "bn.weight"], weights[p + "bn.bias"]
            mu, var = weights[p + "bn.running_mean"], weights[p + "bn.running_var"]
            scale = g / torch.sqrt(var + BN_EPS)
            out[k] = v * scale.view(-1, 1, 1, 1)
            out[p + "conv.bias"] = b - g * mu / torch.sqrt(var + BN_EPS)
        elif ".bn." in k:
            continue
        else:
            out[k] = v
    return out


# ----------------------------------------------------------------------------------------------
# functional forward
# ----------------------------------------------------------------------------------------------
class _Ctx:
    def __init__(self, weights, training, fused, bn_updates, act_round=None):
        self.w, self.training, self.fused, self.bn_updates = weights, training, fused, bn_updates
        self.act_round = act_round  # optional callable emulating reduced-precision activation storage


def _conv_unit(ctx: _Ctx, x, prefix, k, s):
    """Conv = SiLU(BN(conv(x))) (common.py:61-62) or SiLU(conv(x)+b) when fused (common.py:64-65)."""
    w = ctx.w[f"{prefix}.conv.weight"]
    if ctx.fused:
        y = F.conv2d(x, w, ctx.w[f"{prefix}.conv.bias"], stride=s, padding=k // 2)
    else:
        y = F.conv2d(x, w, None, stride=s, padding=k // 2)
        if ctx.act_round is not None:
            y = ctx.act_round(y)
        g, b = ctx.w[f"{prefix}.bn.weight"], ctx.w[f"{prefix}.bn.bias"]
        if ctx.training:
            mean = y.mean(dim=(0, 2, 3))
            var = y.var(dim=(0, 2, 3), unbiased=False)
            if ctx.bn_updates is not None:
                n = y.numel() / y.shape[1]
                ctx.bn_updates[f"{prefix}.bn.running_mean"] = (
                    (1 - BN_MOMENTUM) * ctx.w[f"{prefix}.bn.running_mean"] + BN_MOMENTUM * mean.detach()
                )
                ctx.bn_updates[f"{prefix}.bn.running_var"] = (
                    (1 - BN_MOMENTUM) * ctx.w[f"{prefix}.bn.running_var"]
                    + BN_MOMENTUM * var.detach() * (n / max(n - 1, 1))
                )
        else:
            mean, var = ctx.w[f"{prefix}.bn.running_mean"], ctx.w[f"{prefix}.bn.running_var"]
        y = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
        y = y * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    y = y * torch.sigmoid(y)
    if ctx.act_round is not None:
        y = ctx.act_round(y)
    return y


def _c2f(ctx, x, spec, prefix):  # common.py:187-191
    c = int(spec["c2"] * 0.5)
    y = _conv_unit(ctx, x, f"{prefix}.cv1", 1, 1)
    ys = [y[:, :c], y[:, c:]]
    for i in range(spec["n"]):
        t = _conv_unit(ctx, ys[-1], f"{prefix}.m.{i}.cv1", 3, 1)
        t = _conv_unit(ctx, t, f"{prefix}.m.{i}.cv2", 3, 1)
        if spec["shortcut"]:  # Bottleneck.add: shortcut and c1 == c2 (always equal inside C2f), common.py:114
            t = ys[-1] + t
            if ctx.act_round is not None:
                t = ctx.act_round(t)
        ys.append(t)
    return _conv_unit(ctx, torch.cat(ys, 1), f"{prefix}.cv2", 1, 1)


def _sppf(ctx, x, spec, prefix):  # common.py:239-245
    k = spec["k"]
    x = _conv_unit(ctx, x, f"{prefix}.cv1", 1, 1)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return _conv_unit(ctx, torch.cat([x, y1, y2, y3], 1), f"{prefix}.cv2", 1, 1)


def _apply(ctx, spec, prefix, xs):
    op = spec["op"]
    if op == "Conv":
        return _conv_unit(ctx, xs[0], prefix, spec["k"], spec["s"])
    if op == "C2f":
        return _c2f(ctx, xs[0], spec, prefix)
    if op == "SPPF":
        return _sppf(ctx, xs[0], spec, prefix)
    if op == "Upsample":
        return F.interpolate(xs[0], scale_factor=2, mode="nearest")
    if op == "Concat":
        return torch.cat(xs, 1)
    raise NotImplementedError(op)


def make_anchors(shapes, strides, offset=0.5, dtype=torch.float32):
    """utils/tal.py:181-193. shapes = [(h, w), ...] -> ([A,2] anchor points, [A,1] stride)."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=dtype) + offset
        sy = torch.arange(h, dtype=dtype) + offset
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=dtype))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):  # utils/tal.py:196-205
    lt, rb = torch.split(distance, 2, dim)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


def detect_decode(feats, nc, strides):
    """Eval branch of Detect.forward (yolo.py:93-100) incl. DFL (yolo.py:57-59): -> y [N, 4+nc, A]."""
    n = feats[0].shape[0]
    no = nc + 4 * REG_MAX
    x = torch.cat([f.reshape(n, no, -1) for f in feats], 2)
    box, cls = x.split((4 * REG_MAX, nc), 1)
    a = box.shape[-1]
    prob = box.view(n, 4, REG_MAX, a).softmax(2)
    proj = torch.arange(REG_MAX, dtype=box.dtype).view(1, 1, REG_MAX, 1)
    dist = (prob * proj).sum(2)  # [N,4,A]
    anchors, st = make_anchors([f.shape[2:] for f in feats], strides, dtype=box.dtype)
    dbox = dist2bbox(dist, anchors.t().unsqueeze(0), xywh=True, dim=1) * st.t()
    return torch.cat((dbox, cls.sigmoid()), 1)


def _detect(ctx, node, xs, strides):
    prefix = node["prefix"]
    feats = []
    for lvl, x in enumerate(xs):
        a = _conv_unit(ctx, x, f"{prefix}.cv2.{lvl}.0", 3, 1)
        a = _conv_unit(ctx, a, f"{prefix}.cv2.{lvl}.1", 3, 1)
        a = F.conv2d(a, ctx.w[f"{prefix}.cv2.{lvl}.2.weight"], ctx.w[f"{prefix}.cv2.{lvl}.2.bias"])
        b = _conv_unit(ctx, x, f"{prefix}.cv3.{lvl}.0", 3, 1)
        b = _conv_unit(ctx, b, f"{prefix}.cv3.{lvl}.1", 3, 1)
        b = F.conv2d(b, ctx.w[f"{prefix}.cv3.{lvl}.2.weight"], ctx.w[f"{prefix}.cv3.{lvl}.2.bias"])
        feats.append(torch.cat((a, b), 1))
    if ctx.training:
        return feats
    return detect_decode(feats, node["nc"], strides), feats


def forward(graph, weights, x, task_ids=None, training=True, fused=False, strides=(8.0, 16.0, 32.0),
            bn_updates=None, act_round=None, return_blocks=False):
    """CerberusDet.forward (cerberus.py:804-882): str task -> that head's output, else {task: output}."""
    ctx = _Ctx(weights, training, fused, bn_updates, act_round)
    tasks = list(graph["heads"].keys()) if task_ids is None else task_ids
    plan = execution_plan(graph, tasks)
    outs: Dict[int, object] = {}
    results = {}
    for idx in plan:
        node = graph["nodes"][idx]
        if node["op"] == "Backbone":
            ys: List[torch.Tensor] = []
            cur = x
            for spec in node["layers"]:
                f = spec["f"]
                if f == -1:
                    xin = [cur]
                elif isinstance(f, int):
                    xin = [ys[f]]
                else:
                    xin = [cur if j == -1 else ys[j] for j in f]
                cur = _apply(ctx, spec, spec["prefix"], xin)
                ys.append(cur)
            outs[0] = ys
            continue
        xs = [outs[0][j] if kind == "bb" else outs[j] for kind, j in node["inputs"]]
        if node["op"] == "Detect":
            outs[idx] = _detect(ctx, node, xs, strides)
            results[node["task"]] = outs[idx]
        else:
            outs[idx] = _apply(ctx, node, node["prefix"], xs)
    if return_blocks:
        return results, outs
    return results[task_ids] if isinstance(task_ids, str) else results


# ----------------------------------------------------------------------------------------------
# accounting (README.md:235-243 known answers)
# ----------------------------------------------------------------------------------------------
def conv_flops_and_params(graph, imgsz=640, tasks=None):
    """2*MACs over every nn.Conv2d executed for `tasks` at imgsz, and parameter count of the
    whole model (SURVEY.md section 8d 'algorithmic work')."""
    shapes = param_shapes(graph)
    n_params = sum(int(np.prod(s)) for k, s in shapes.items()
                   if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    flops = [0]

    def conv(hw, c1, c2, k, s):
        ho = (hw + 2 * (k // 2) - k) // s + 1
        flops[0] += 2 * c2 * ho * ho * c1 * k * k
        return ho

    def unit(spec, hw):
        op = spec["op"]
        if op == "Conv":
            return conv(hw, spec["c1"], spec["c2"], spec["k"], spec["s"])
        if op == "C2f":
            c = int(spec["c2"] * 0.5)
            conv(hw, spec["c1"], 2 * c, 1, 1)
            for _ in range(spec["n"]):
                conv(hw, c, c, 3, 1)
                conv(hw, c, c, 3, 1)
            conv(hw, (2 + spec["n"]) * c, spec["c2"], 1, 1)
            return hw
        if op == "SPPF":
            conv(hw, spec["c1"], spec["c1"] // 2, 1, 1)
            conv(hw, spec["c1"] // 2 * 4, spec["c2"], 1, 1)
            return hw
        if op == "Upsample":
            return hw * 2
        return hw

    tasks = list(graph["heads"].keys()) if tasks is None else tasks
    plan = execution_plan(graph, tasks)
    hw_of: Dict[object, int] = {}
    for idx in plan:
        node = graph["nodes"][idx]
        if node["op"] == "Backbone":
            hw = imgsz
            for j, spec in enumerate(node["layers"]):
                hw = unit(spec, hw)
                hw_of[("bb", j)] = hw
            continue
        hws = [hw_of[(k, j)] for k, j in node["inputs"]]
        if node["op"] == "Detect":
            nc, ch = node["nc"], node["ch"]
            c2, c3 = max(16, ch[0] // 4, REG_MAX * 4), max(ch[0], nc)
            for x, hw in zip(ch, hws):
                conv(hw, x, c2, 3, 1); conv(hw, c2, c2, 3, 1); conv(hw, c2, 4 * REG_MAX, 1, 1)
                conv(hw, x, c3, 3, 1); conv(hw, c3, c3, 3, 1); conv(hw, c3, nc, 1, 1)
                flops[0] += 2 * 4 * hw * hw * REG_MAX  # DFL 16->1 conv on [4, A]
        else:
            hw_of[("blk", idx)] = unit(node, hws[0])
    return flops[0], n_params
