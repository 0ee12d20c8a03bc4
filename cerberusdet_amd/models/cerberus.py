"""CerberusDet: shared backbone + per-task neck branches + one Detect head per task, as a block DAG.

Drop-in for the reference's `cerberusdet.models.cerberus.CerberusDet` (ai-forever/CerberusDet
cerberusdet/models/cerberus.py): same constructor, attributes (`blocks`, `controllers`, `heads`, `stride`, `yaml`,
`branching_points`, `neck_head_save`), methods (`forward`, `execution_plan`, `control_blocks`, `parameters`, `get_head`,
`sequential_split`, `split`, `create_nested_branch`, `fuse`, `info`, `freeze_/unfreeze_shared_layers`), block numbering
and state-dict keys. What differs is execution: `forward` compiles the DAG once into a static launch list of hand-written
gfx950 kernels (engine.Plan) instead of dispatching torch ops block by block (reference cerberus.py:804-882).
"""
from __future__ import annotations

import itertools
import os
from collections import defaultdict
from copy import deepcopy
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from .common import Conv
from .yolo import Detect, Model, get_next_layer_from_cfg, initialize_weights

LOCAL_RANK = int(os.getenv("LOCAL_RANK", -1))


class Controller:
    """Node of the block DAG (reference cerberus.py:38-139): index, execution_chain (every block that must run before
    this one, then the block itself), parent(s), children, task_id for heads, serving_tasks."""

    def __init__(self, index=None):
        self.index = index
        self.execution_chain = [index]
        self.parent_index = None
        self.children_indices = []
        self.task_id = None
        self.serving_tasks = dict()

    def __str__(self):
        return "({}): parent={}, children={}, serving=[{}]".format(
            self.index, self.parent_index, self.children_indices, ", ".join(str(t) for t in self.serving_tasks))

    __repr__ = __str__

    def serialize(self):
        return self.__dict__

    def deserialize(self, d):
        for k, v in d.items():
            setattr(self, k, v)
        return self


class CerberusDet(nn.Module):
    def __init__(self, task_ids, nc, cfg="v8x_2task.yaml", ch=3, verbose=True, **kwargs):
        super().__init__()
        self.blocks = nn.ModuleList()
        self.controllers: List[Controller] = []
        self.heads: Dict[str, int] = dict()
        self.rep_tensors = dict()
        self.branching_points = set()
        self.verbose = verbose
        self._inputs: Dict[int, list] = {}     # block idx -> [("blk", j) | ("bb", backbone_layer)]
        self._plans = {}
        self._pgrad = {}                       # id(param) -> persistent fp32 gradient buffer the kernels accumulate into
        self._weights_version = 0
        self.compute_dtype = torch.bfloat16    # storage dtype of activations / packed weights (bf16 or fp16)

        nc = [nc] * len(task_ids) if isinstance(nc, int) else list(nc)
        backbone = Model(cfg=cfg, ch=ch, nc=list(nc), without_head=True, verbose=verbose, **kwargs)
        self._add_block(backbone)
        self.gd, self.gw = backbone.yaml["depth_multiple"], backbone.yaml["width_multiple"]
        self.max_channels = backbone.yaml.get("max_channels", 1024)
        cfgd = deepcopy(backbone.yaml)
        nb = len(cfgd["backbone"])
        chs = list(backbone.saved_ch)
        self.neck_head_save: List[int] = []

        # ---- neck: one block per YAML row (reference cerberus.py:212-254)
        layer_ind_map = {}
        prev = 0
        for i, (f, n, m, args) in enumerate(cfgd["neck"], start=1):
            ind = nb + i - 1
            _, _, _, c2, m_ = get_next_layer_from_cfg(self.gd, chs, self.gw, list(nc), m, n, f, args, self.max_channels)
            chs.append(c2)
            fl = [f] if isinstance(f, int) else list(f)
            assert fl[0] == -1 or len(fl) == 1, "Unsupported config"
            idx = self._add_block(m_)
            ins, new_f = [], []
            for x in fl:
                if x == -1:
                    if i == 1:
                        raise ValueError("Input for first cerbernet block must be defined")
                    ins.append(("blk", prev))
                elif x >= nb:
                    j = layer_ind_map[x]
                    self.neck_head_save.append(j)
                    ins.append(("blk", j))
                    x = j
                else:
                    ins.append(("bb", x))
                    x = (0, x)
                new_f.append(x)
            self._inputs[idx] = ins
            m_.i, m_.f, m_.type = ind, new_f, (m if isinstance(m, str) else m.__name__)
            m_.np = sum(p.numel() for p in m_.parameters())
            layer_ind_map[ind] = idx
            prev = idx

        # ---- heads: one Detect per task (reference cerberus.py:256-319)
        if len(cfgd["head"]) != 1:
            raise NotImplementedError
        f, n, m, args = cfgd["head"][0]
        nc_left = list(nc)
        ind = nb + len(cfgd["neck"])
        for task_id in task_ids:
            _, nc_left, _, _, m_ = get_next_layer_from_cfg(self.gd, chs, self.gw, nc_left, m, n, f, deepcopy(args), self.max_channels)
            idx = self._add_block(m_)
            self.controllers[idx].task_id = task_id
            self.heads[task_id] = idx
            ins, new_f = [], []
            for x in f:
                if x < nb:
                    raise ValueError("Input for the head must be from neck")
                j = layer_ind_map[x]
                self.neck_head_save.append(j)
                ins.append(("blk", j))
                new_f.append(j)
            self._inputs[idx] = ins
            m_.i, m_.f, m_.type = ind, new_f, "Detect"
            m_.np = sum(p.numel() for p in m_.parameters())
            m_.inplace = backbone.inplace
        del backbone.saved_ch
        for block in self.blocks:
            initialize_weights(block)
        self.yaml = backbone.yaml
        self.build()
        # strides: the reference probes with a 256x256 zero forward (cerberus.py:299-316); the geometry is static, so
        # derive it from the graph instead.
        stride = kwargs.get("stride") or self._infer_strides(256)
        self.stride = torch.tensor(stride, dtype=torch.float32)
        for t in task_ids:
            h = self.get_head(t)
            h.stride = torch.tensor(stride, dtype=torch.float32)
            h.bias_init()

    # ------------------------------------------------------------------------------------------------------ graph
    def _add_block(self, module) -> int:
        idx = len(self.blocks)
        self.blocks.append(module)
        self.controllers.append(Controller(idx))
        return idx

    def add_block(self, module):
        return self.controllers[self._add_block(module)]

    def add_head(self, module, task_id):
        c = self.add_block(module)
        c.task_id = task_id
        self.heads[task_id] = c.index
        return c

    def _parents(self, idx) -> List[int]:
        out = []
        for kind, j in self._inputs.get(idx, []):
            j = 0 if kind == "bb" else j
            if j not in out:
                out.append(j)
        return out

    def _ancestors(self, idx, acc=None):
        acc = set() if acc is None else acc
        for j in self._parents(idx):
            if j not in acc:
                acc.add(j)
                self._ancestors(j, acc)
        return acc

    def build(self):
        """(Re)derive controllers from the DAG: chains, parents/children, serving_tasks, branching points
        (reference cerberus.py:449-459)."""
        for c in self.controllers:
            c.children_indices = []
            c.serving_tasks = dict()
        for c in self.controllers:
            par = self._parents(c.index)
            c.parent_index = None if not par else (par[0] if len(par) == 1 else par)
            c.execution_chain = sorted(self._ancestors(c.index)) + [c.index]
            for j in par:
                self.controllers[j].children_indices.append(c.index)
        for _, head_index in self.heads.items():
            controller = self.controllers[head_index]
            for index in controller.execution_chain:
                self.controllers[index].serving_tasks[controller.task_id] = len(self.controllers[index].serving_tasks)
        _, self.branching_points = self.execution_plan(list(self.heads.keys()))
        self._plans = {}

    def execution_plan(self, task_ids: Union[List[str], str]):
        """Order of blocks for the given task(s) and the branching points (reference cerberus.py:371-403)."""
        if not isinstance(task_ids, list):
            task_ids = [task_ids]
        execution_order: List[int] = []
        branching_ids = set()
        for task_id in task_ids:
            chain = self.controllers[self.heads[task_id]].execution_chain
            branching_point = None
            i = 0
            for i, index in enumerate(chain):
                if index not in execution_order:
                    break
                branching_point = index
            execution_order += chain[i:]
            if branching_point is not None:
                parents = self.controllers[chain[i]].parent_index
                if isinstance(parents, int):
                    branching_ids.add(branching_point)
                else:
                    branching_ids.update(p for p in parents if p in execution_order)
        return execution_order, branching_ids

    def _infer_strides(self, s=256):
        """Spatial size of every head input for an s x s image -> strides (replaces the reference's probe forward)."""
        def conv_out(h, m):
            return (h + 2 * (m.k // 2) - m.k) // m.s + 1

        from .common import C2f, Concat, SPPF, Upsample

        def step(m, hs):
            if isinstance(m, Conv):
                return conv_out(hs[0], m)
            if isinstance(m, Upsample):
                return hs[0] * 2
            return hs[0]

        bb = []
        h = s
        for layer in self.blocks[0].model:
            f = layer.f
            hin = [h] if f == -1 else ([bb[f]] if isinstance(f, int) else [h if j == -1 else bb[j] for j in f])
            h = step(layer, hin)
            bb.append(h)
        outs = {}
        order, _ = self.execution_plan(list(self.heads.keys()))
        strides = None
        for idx in order:
            if idx == 0:
                continue
            hin = [bb[j] if kind == "bb" else outs[j] for kind, j in self._inputs[idx]]
            if idx in self.heads.values():
                st = [s / x for x in hin]
                assert strides is None or strides == st
                strides = st
            else:
                outs[idx] = step(self.blocks[idx], hin)
        return strides

    def get_head(self, task_id) -> Detect:
        return self.blocks[self.heads[task_id]]

    def info(self):
        items = "\n  ".join(str(c) for c in self.controllers)
        heads = "\n  ".join("({}) -> {}  {}".format(k, c, type(self.blocks[c])) for k, c in self.heads.items())
        return "(block controllers):\n  " + items + "\n(heads):\n  " + heads

    def control_blocks(self, task_ids=None):
        if task_ids is None:
            for controller, block in zip(self.controllers, self.blocks):
                yield controller, block
        else:
            order, _ = self.execution_plan(task_ids)
            for index in order:
                yield self.controllers[index], self.blocks[index]

    def parameters(self, recurse=True, task_ids=None, only_trainable=False):
        if task_ids is None and not only_trainable:
            yield from super().parameters(recurse)
            return
        if task_ids is None:
            task_ids = list(self.heads.keys())
        order, _ = self.execution_plan(task_ids)
        for index in order:
            if only_trainable and getattr(self.blocks[index], "trainable", None) is not True:
                continue
            yield from self.blocks[index].parameters()

    # ------------------------------------------------------------------------------------------------------ branching
    def create_nested_branch(self, index: int, branches: List[int], device=None, inds_to_map_per_head=None, next_ids_map=None):
        """Clone every block after `index` on the path of the heads in `branches` and re-wire those heads onto the clones
        (reference cerberus.py:461-633). Clones are appended at the end of `self.blocks` in execution order."""
        if index in self.heads.values():
            raise ValueError("Cannot split 's head.")
        names = [t for t, i in self.heads.items() if i in branches]
        if len(names) != len(branches):
            raise ValueError("Indices of branches must be indexes of heads.")
        order, _ = self.execution_plan(names)
        clones: Dict[int, int] = {}
        new_ctrl, new_blocks = [], []
        for ind in order:
            if ind <= index:
                continue
            if ind in branches:
                break
            src_block = self.blocks[ind]
            stash = {}
            for mod in src_block.modules():
                st = {a: mod.__dict__.pop(a) for a in self._RUNTIME_ATTRS if a in mod.__dict__}
                if st:
                    stash[mod] = st
            block = deepcopy(src_block)
            for mod, st in stash.items():
                mod.__dict__.update(st)
            if device is not None:
                block = block.to(device)
            new = self._add_block(block)
            clones[ind] = new
            self._inputs[new] = [(k, clones.get(j, j)) if k == "blk" else (k, j) for k, j in self._inputs[ind]]
            block.f = [clones.get(x, x) if isinstance(x, int) and x != -1 else x for x in list(block.f)]
            if ind in self.neck_head_save:
                self.neck_head_save.append(new)
            new_ctrl.append(self.controllers[new])
            new_blocks.append(block)
        for h in branches:
            self._inputs[h] = [(k, clones.get(j, j)) if k == "blk" else (k, j) for k, j in self._inputs[h]]
            self.blocks[h].f = [clones.get(x, x) if isinstance(x, int) and x != -1 else x for x in list(self.blocks[h].f)]
        self.rep_tensors.clear()
        self.build()
        if inds_to_map_per_head is not None:
            for old, new in clones.items():
                for h in branches:
                    if h in inds_to_map_per_head and old in inds_to_map_per_head[h]:
                        next_ids_map[h][old] = new
        return new_ctrl, new_blocks

    def split(self, index, branching_scheme, device, next_cerber_configs):
        """First group keeps the blocks after `index`, each other group gets its own copies (reference cerberus.py:635-702)."""
        inds_to_map_per_head = defaultdict(list)
        next_ids_map: Dict[int, Dict[int, Optional[int]]] = {}
        for sc in next_cerber_configs:
            for head_ind in itertools.chain(*sc[1]):
                inds_to_map_per_head[head_ind].append(sc[0])
                next_ids_map[head_ind] = {sc[0]: None}
                if head_ind in branching_scheme[0]:
                    next_ids_map[head_ind][sc[0]] = sc[0]
        for a in range(len(branching_scheme)):
            for b in range(a + 1, len(branching_scheme)):
                if not set(branching_scheme[a]).isdisjoint(branching_scheme[b]):
                    raise ValueError("The branching schemes should be disjoint to each other.")
        ctrls, blocks = [self.controllers[index]], [self.blocks[index]]
        for branch in branching_scheme[1:]:
            c, b = self.create_nested_branch(index, branch, device, inds_to_map_per_head, next_ids_map)
            ctrls.append(c)
            blocks.append(b)
        return ctrls, blocks, next_ids_map

    def sequential_split(self, cerber_schedule, device):
        """Apply the YAML `cerber` schedule [[neck_idx, [[heads..], [heads..]]], ...] (reference cerberus.py:704-737)."""
        cerber_schedule = deepcopy(cerber_schedule)
        sched_heads = sorted({h for conf in cerber_schedule for h in itertools.chain(*conf[-1])})
        assert sorted(self.heads.values()) == sched_heads or len(sched_heads) == 0, f"Invalid cerberusNet config {cerber_schedule}"
        for i in range(len(cerber_schedule)):
            nxt = cerber_schedule[i + 1:]
            _, _, ids_map = self.split(*cerber_schedule[i], device, nxt)
            for ii, sc in enumerate(nxt):
                mapped = [ids_map[h][sc[0]] for h in itertools.chain(*sc[1])]
                assert None not in mapped and len(set(mapped)) == 1
                cerber_schedule[i + 1 + ii][0] = mapped[0]

    # ------------------------------------------------------------------------------------------------------ misc API
    def fuse(self):
        """Fold BatchNorm into the convolutions for inference (reference cerberus.py:739-757)."""
        for m in self.modules():
            if type(m) is Conv:
                m.fuse_()
        self.mark_weights_changed()
        self._drop_plans()
        return self

    def mark_weights_changed(self):
        """Call after modifying parameters in place (optimizer step, load_state_dict): packed bf16 copies are refreshed lazily."""
        self._weights_version += 1

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.mark_weights_changed()
        return out

    def state_dict(self, *a, **k):
        self._flush_bn_counters()
        return super().state_dict(*a, **k)

    _RUNTIME_ATTRS = ("_plan_slots", "_pack_key", "_wp", "_wpt", "_scale", "_bias", "_bias_pad", "_w8", "_gw8", "_stem8", "_wp_zeroed",
                      "_wt_f", "_wt_d", "_ws1")

    def _grad_buffer(self, p):
        g = self._pgrad.get(id(p))
        if g is None or g.device != p.device or g.shape != p.shape:
            g = torch.zeros_like(p, dtype=torch.float32)
            self._pgrad[id(p)] = g
        return g

    def _merge_alt_grads(self):
        """Fold the per-task gradient buckets of the shared blocks (installed by trainers.Averaging; engine.Plan._alt_grad) into the blocks' own
        buckets, in task order, on the current stream, and clear them. No-op for a model without such buckets."""
        pairs = getattr(self, "_alt_pairs", None)
        if not pairs:
            return
        from .. import _lib as L

        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        for main, alt in pairs:
            L.check(lib.cdet_accumulate_clear(main.data_ptr(), alt.data_ptr(), main.numel(), st), "cdet_accumulate_clear")

    def _autograd_anchor(self):
        a = getattr(self, "_anchor", None)
        if a is None or a.device != next(super().parameters()).device:
            a = torch.zeros(1, device=next(super().parameters()).device, requires_grad=True)
            object.__setattr__(self, "_anchor", a)
        return a

    def __deepcopy__(self, memo):
        """deepcopy (ModelEMA, branch cloning) must not drag compiled plans / ctypes descriptors along."""
        saved = {}
        for m in self.modules():
            st = {a: m.__dict__.pop(a) for a in self._RUNTIME_ATTRS if a in m.__dict__}
            if st:
                saved[m] = st
        plans, pgrad, anchor = self._plans, self._pgrad, self.__dict__.pop("_anchor", None)
        peer = self.__dict__.pop("_peer_xchg", None)  # (IPC-mapped exchange buffers belong to the trainer's model, not to its EMA copy)
        alt = (self.__dict__.pop("_pgrad_alt", None), self.__dict__.pop("_alt_pairs", None))  # (the trainer's per-task gradient buckets likewise)
        self._plans, self._pgrad = {}, {}
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = deepcopy(v, memo)
        finally:
            self._plans, self._pgrad = plans, pgrad
            if peer is not None:
                self._peer_xchg = peer
            if alt[0] is not None:
                self._pgrad_alt, self._alt_pairs = alt
            if anchor is not None:
                object.__setattr__(self, "_anchor", anchor)
            for m, st in saved.items():
                m.__dict__.update(st)
        return new

    def _flush_bn_counters(self):
        for m in self.modules():
            n = getattr(m, "_nbt_pending", 0)
            if n and isinstance(m, nn.BatchNorm2d):
                m.num_batches_tracked += n
                m._nbt_pending = 0

    def _drop_plans(self):
        """Forget every compiled plan. Plans register their argument lists on the modules (`_plan_slots`, re-bound by every weight pack): release()
        unregisters them -- dropping the dict alone left dead (kind, slot) entries behind for every dtype switch."""
        for plan in list(getattr(self, "_plans", {}).values()):
            rel = getattr(plan, "release", None)
            if rel is not None:
                rel()
        self._plans = {}

    def _apply(self, fn, *a, **k):
        self._drop_plans()  # device / dtype moves invalidate every pre-bound pointer
        self._pgrad = {}
        self.__dict__.pop("_anchor", None)
        for m in self.modules():
            for attr in self._RUNTIME_ATTRS:
                if attr in m.__dict__:
                    del m.__dict__[attr]
        out = super()._apply(fn, *a, **k)
        self._weights_version += 1
        return out

    def half(self):
        """Reference inference calls model.half() (cerberusdet_inference.py:39-40): parameters stay fp32 masters here,
        only the compute/storage dtype of activations and packed weights switches to fp16."""
        if self.compute_dtype != torch.float16:
            self.compute_dtype = torch.float16
            self._drop_plans()
        return self

    def bfloat16(self):
        if self.compute_dtype != torch.bfloat16:
            self.compute_dtype = torch.bfloat16
            self._drop_plans()
        return self

    def float(self):
        """No-op: parameters are fp32 masters already; the compute dtype is chosen with half() / bfloat16() / full_precision()."""
        return self

    def full_precision(self):
        """Evaluate at the reference's own precision -- what `model.float()` on fp32 inputs gives there (cerberus.py:804-882): every convolution as
        six bf16 term-pair launches of the product's MFMA kernels accumulated in fp32, everything between them in fp32 (cerberusdet_amd/precise.py).
        Boxes / maps agree with the fp32 reference to <= 1e-3 (BASELINE.json's tolerance); roughly 8x the time of the bf16 plan. Eval mode, or train mode
        (batch-statistics BatchNorm, running statistics updated) with a backward at the same precision through autograd: `model(x, task)` ->
        `loss.backward()` accumulates into the parameters' fp32 `.grad`; trainers.Averaging runs such a model as sequential task passes."""
        if self.compute_dtype != torch.float32:
            self.compute_dtype = torch.float32
            self._drop_plans()
        return self

    def set_task(self, task_id):
        self.cur_task = task_id

    def test_forward(self, device=None):
        x = torch.ones(1, 3, 256, 256, device=device or next(self.parameters()).device)
        self.forward(x)

    # ------------------------------------------------------------------------------------------------------ execution
    def get_plan(self, task_ids, shape, img_dtype, training=None):
        from ..engine import Plan

        training = self.training if training is None else training
        tasks = [task_ids] if isinstance(task_ids, str) else list(task_ids)
        if self.compute_dtype == torch.float32:
            raise NotImplementedError("compiled launch-list plans (engine.Plan: what trainers.Averaging drives) store activations in 16 bits: a "
                                      "full_precision() model runs through model(x) / autograd only -- call model.bfloat16() / model.half() first")
        frozen = ()
        if training:  # blocks whose parameters are all frozen (freeze_shared_layers): train-form forward with batch statistics, running
            # statistics untouched, no backward (engine.Plan.frozen)
            frozen = tuple(i for i, b in enumerate(self.blocks)
                           if any(True for _ in b.parameters()) and not any(p.requires_grad for p in b.parameters()))
        key = (tuple(tasks), tuple(shape), img_dtype, training, self.compute_dtype, bool(getattr(self, "sync_bn", False)), frozen)
        plan = self._plans.pop(key, None)
        if plan is None:
            dev = next(super().parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("cerberusdet_amd runs on an MI355X only: move the model to 'cuda' (there is no CPU path)")
            N, c, H, W = shape
            if not training:
                # every plan owns its activation buffers (a YOLOv8x eval plan at batch 32 @640: ~6 GB): rectangular validation batches
                # (one frame shape per aspect-ratio bucket, data.rect_batch_shapes) or odd last batches would otherwise pile up one
                # resident plan per shape. Least recently used eval plans beyond CDET_MAX_EVAL_PLANS (default 6) are dropped.
                cap = int(os.environ.get("CDET_MAX_EVAL_PLANS", "6"))
                evals = [k for k in self._plans if k[3] is False]
                for k in evals[:max(len(evals) - cap + 1, 0)]:
                    self._plans.pop(k).release()
            plan = Plan(self, tasks, N, H, W, training, self.compute_dtype, img_dtype, dev, frozen=frozen)
        self._plans[key] = plan  # (re-inserted last: dict order = recency)
        return plan

    def forward(self, input_tensor, task_ids=None, retain_tensors=False, retain_all=False, zero_copy=False, contiguous_maps=False):
        """Same contract as the reference (cerberus.py:804-882): a `str` task -> that head's output, otherwise a dict.
        train mode -> list of 3 raw maps [N, 64+nc, h, w]; eval mode -> (y [N, 4+nc, A], maps).
        Like the reference, every call returns FRESH tensors. Eval mode: the head projections and the decode kernel write straight
        into a newly allocated output set (engine.Plan.fresh_outputs: pointer patches on the host, no device copy). Train mode: one
        flat device copy of the padded NHWC head maps (~0.1 ms at batch 32 @640; the loss kernels keep reading the plan's own maps),
        returned as NCHW-shaped views.
        `zero_copy=True` returns VIEWS of the compiled plan's buffers instead: the next forward of the same (tasks, shape, dtype, mode)
        overwrites them, so consume them (NMS, loss) before calling the model again -- CerberusDetInference, val.run and the trainer
        do. Either way a train-mode forward keeps ONE set of saved activations per configuration: call backward() before the next
        forward of the same configuration (autograd_bridge refuses a stale backward)."""
        if retain_tensors or retain_all:
            # the reference stores block outputs in self.rep_tensors for its branching-analysis tooling (cerberus.py:866-872); the compiled
            # launch list keeps no per-block NCHW tensors (Concat / Upsample outputs do not even exist in eval plans)
            raise NotImplementedError("retain_tensors / retain_all (rep_tensors of the reference's branching analysis) are not supported by cerberusdet_amd")
        if task_ids is None and hasattr(self, "cur_task"):
            task_ids = self.cur_task
        elif task_ids is None:
            task_ids = list(self.heads.keys())
        tasks = [task_ids] if isinstance(task_ids, str) else list(task_ids)
        x = input_tensor.contiguous()
        if self.compute_dtype == torch.float32:
            return self._forward_full_precision(tasks, x, task_ids, zero_copy)
        plan = self.get_plan(tasks, x.shape, x.dtype)
        if self.training and torch.is_grad_enabled():
            from ..autograd_bridge import run_with_autograd

            outs = run_with_autograd(plan, x, fresh=not zero_copy)
        else:
            if not self.training:
                plan.fresh_outputs(not zero_copy)
            plan.run_forward(x)
            outs = {}
            for t in tasks:
                nc = self.get_head(t).nc
                srcs = [f.clone() for f in plan.feats[t]] if (self.training and not zero_copy) else plan.feats[t]  # flat copies
                # NCHW-shaped VIEWS of the padded NHWC fp32 maps (values and shapes as the reference's Detect.forward returns them,
                # models/yolo.py:87-100; `.view()` needs contiguous memory: contiguous_maps=True pays one transposing copy per map for it)
                maps = [f[..., :64 + nc].permute(0, 3, 1, 2) for f in srcs]
                if contiguous_maps:
                    maps = [m.contiguous() for m in maps]
                outs[t] = maps if self.training else (plan.y[t], maps)
        return outs[task_ids] if isinstance(task_ids, str) else outs

    def full_precision_plan(self, tasks, x, training: bool):
        """The cached PrecisePlan of (tasks, image shape / dtype, mode) of a full_precision() model."""
        from ..precise import PrecisePlan

        # (train plans are compiled for the CURRENT set of trainable parameters -- frozen ones get no gradient and cut the data-gradient chain,
        #  precise.PrecisePlan -- so the mask is part of the key: freeze_shared_layers / unfreeze_shared_layers select different plans)
        mask = hash(tuple(p.requires_grad for p in super().parameters())) if training else 0
        key = (tuple(tasks), tuple(x.shape), x.dtype, False, torch.float32, "full_precision", mask, bool(training))
        plan = self._plans.pop(key, None)
        if plan is None:
            cap = int(os.environ.get("CDET_MAX_EVAL_PLANS", "6"))
            evals = [k for k in self._plans if k[3] is False]
            for k in evals[:max(len(evals) - cap + 1, 0)]:
                self._plans.pop(k).release()
            plan = PrecisePlan(self, tasks, x.shape[0], x.shape[2], x.shape[3], x.dtype, next(super().parameters()).device, training=bool(training))
        self._plans[key] = plan
        return plan

    def _forward_full_precision(self, tasks, x, task_ids, zero_copy=False):
        """Forward of a full_precision() model (cerberusdet_amd/precise.py). Like the 16-bit plans: fresh tensors per call (one flat copy of the head
        maps) unless zero_copy=True, which returns views of the plan's maps that the next forward of the configuration overwrites."""
        dev = next(super().parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("cerberusdet_amd runs on an MI355X only: move the model to 'cuda' (there is no CPU path)")
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16, torch.uint8):
            raise TypeError(f"full_precision(): image dtype {x.dtype} (expected uint8, divided by 255 like the reference's preprocess_batch, or a "
                            "floating-point image already scaled to [0, 1])")
        plan = self.full_precision_plan(tasks, x, self.training)
        if self.training and torch.is_grad_enabled():
            from ..precise import run_with_autograd

            outs = run_with_autograd(plan, x)
            return outs[task_ids] if isinstance(task_ids, str) else outs
        with torch.no_grad():
            plan.run(x)
        outs = {}
        for t in tasks:
            nc = self.get_head(t).nc
            maps = [(f if zero_copy else f.clone())[..., :64 + nc].permute(0, 3, 1, 2) for f in plan.feats[t]]
            outs[t] = maps if self.training else (plan.y[t], maps)
        return outs[task_ids] if isinstance(task_ids, str) else outs

    # ------------------------------------------------------------------------------------------------------ freezing
    @staticmethod
    def freeze_shared_layers(cerberus_model):
        model = cerberus_model.module if hasattr(cerberus_model, "module") else cerberus_model
        if len(model.heads) == 1:
            return
        for ctrl, block in model.control_blocks():
            if max(len(ctrl.serving_tasks), 1) != len(model.heads):
                continue
            for p in block.parameters():
                p.requires_grad = False
            for m in block.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.track_running_stats = False
                    m.eval()

    @staticmethod
    def unfreeze_shared_layers(cerberus_model):
        model = cerberus_model.module if hasattr(cerberus_model, "module") else cerberus_model
        if len(model.heads) == 1:
            return
        for ctrl, block in model.control_blocks():
            if max(len(ctrl.serving_tasks), 1) != len(model.heads):
                continue
            for p in block.parameters():
                p.requires_grad = True
            for m in block.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.track_running_stats = True
                    m.train()
