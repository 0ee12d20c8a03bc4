"""Multi-task "gradient averaging" trainer with the reference's step semantics
(ai-forever/CerberusDet cerberusdet/trainers/averaging.py:97-223, base_trainer.py:100-112, utils/torch_utils.py:282-316):

  per iteration: for every task -> next batch -> forward -> loss -> backward (gradients ACCUMULATE across tasks);
  then: clip the global gradient norm over ALL parameters to 10, divide each block's gradients by the number of tasks it
  serves, SGD-Nesterov step (3 parameter groups), zero the gradients, EMA update.

MI355X-native differences (same algebra):
  * forward/backward are the compiled launch lists of engine.Plan, the criterion is the fused HIP loss;
  * clip + divide + SGD + zero + EMA are TWO kernel launches over a slot table (csrc/optim.hip);
  * data parallelism is explicit: every rank sums its local gradients, `GradReducer` all-reduces (SUM) one flat fp32 bucket
    per block over RCCL as soon as the LAST task that touches the block has finished its backward, overlapped with the rest
    of the backward / the next task's forward. The reference gets the same sum through DDP(avg) * world_size
    (averaging.py:162-163) but reduces every shared parameter once per task pass and all 105 M registered parameters
    (find_unused_parameters) instead of only those on the executed path.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from copy import deepcopy
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib as L
from ..utils.general import one_cycle
from ..utils.loss import pad_targets
from ..utils.torch_utils import ema_decay, get_hyperparameter


def get_param_groups(model):
    """g0 = weights with decay, g1 = BatchNorm weights, g2 = biases (reference averaging.py:249-262)."""
    g0, g1, g2 = [], [], []
    for _, v in model.named_modules():
        if hasattr(v, "bias") and isinstance(v.bias, nn.Parameter):
            g2.append(v.bias)
        if isinstance(v, nn.BatchNorm2d):
            g1.append(v.weight)
        elif hasattr(v, "weight") and isinstance(v.weight, nn.Parameter):
            g0.append(v.weight)
    return g0, g1, g2


class ModelEMA:
    """EMA of every floating-point state-dict entry (reference torch_utils.py:282-316). The lerp itself runs inside the fused
    optimizer kernel; this class owns the shadow model and the update counter."""

    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(model).eval()
        self.updates = updates
        self.decay_base = decay
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def decay(self, x):
        return ema_decay(x, self.decay_base)


class CommTimer:
    """Where a COMPUTE stream waits for communication, measured with event pairs on that stream: the join on the gradient all-reduce handles in
    front of the optimizer step ("grad_wait": what of the 421 MB exchange the backward did not hide) and every SyncBatchNorm statistics
    collective ("syncbn": they sit on the dependent chain of their layer, so their whole duration is exposed). bench.py --gpus N attaches one to
    the model (`model._comm_timer`) and prints the per-iteration sums as `comm_exposed_ms`, so that the first real multi-GPU run explains itself.
    Off (None) everywhere else: two event records per collective are not free."""

    def __init__(self):
        self.pairs: List[tuple] = []

    class _Span:
        def __init__(self, timer, kind):
            self.timer, self.kind = timer, kind

        def __enter__(self):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())

        def __exit__(self, *exc):
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            self.timer.pairs.append((self.kind, self.e0, e1))
            return False

    def span(self, kind: str):
        return CommTimer._Span(self, kind)

    def reset(self):
        self.pairs = []

    def collect(self) -> Dict[str, float]:
        """Milliseconds per kind since the last reset (synchronises)."""
        torch.cuda.synchronize()
        out: Dict[str, float] = {}
        for kind, e0, e1 in self.pairs:
            out[kind] = out.get(kind, 0.0) + e0.elapsed_time(e1)
        out["n_spans"] = float(len(self.pairs))
        return out


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def comm_span(model, kind: str):
    t = getattr(model, "_comm_timer", None)
    return t.span(kind) if t is not None else _NoSpan()


def _block_of(key: str) -> int:
    """Block index of a state-dict key 'blocks.<i>. ...'; -1 (= the optimizer's tail) for anything else, e.g. a floating-point buffer registered on
    the model itself -- such EMA-only slots must not break the trainer's construction."""
    parts = key.split(".")
    return int(parts[1]) if len(parts) > 2 and parts[0] == "blocks" and parts[1].isdigit() else -1


class GradReducer:
    """Bucketed gradient all-reduce keyed on the block DAG. One flat fp32 bucket per block, sent as REDUCTION UNITS: the slices that become
    final at one point of the backward -- every backbone row of block 0, every other block as a whole. A unit is reduced (SUM, async) when the
    last task of the iteration that serves it has produced its gradients. Device-agnostic (works with gloo on CPU tensors for tests, with RCCL
    on the GPU)."""

    def __init__(self, buckets: Dict[int, torch.Tensor], serving: Dict[int, Sequence[str]], task_order: Sequence[str], group=None):
        self.buckets, self.serving, self.task_order, self.group = buckets, serving, list(task_order), group
        # CDET_REDUCE_ALWAYS=1 keeps the collective path on for world_size 1 (single-GPU smoke test of the RCCL plumbing)
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("CDET_REDUCE_ALWAYS") == "1")
        self.handles: List = []
        self.reduced_bytes = 0
        self.skip_blocks = set()  # frozen blocks: no gradient to reduce

    def unit_done(self, u: dict, task: str, active_tasks: Optional[Sequence[str]], fold) -> bool:
        """Sequential task passes (one stream, or CPU tensors under gloo): `task` has just produced its gradients of reduction unit `u`
        (dict(block, main, alts=[(task, bucket)], serving)). Its own bucket of the unit -- every serving task but the first has one -- is folded
        into the block's bucket right here by `fold(main, alt)` (main += alt; alt = 0): passes run in task order, so the sum builds up as
        g_A, + g_B, + g_C, the sum one shared buffer would hold. The unit is all-reduced when this was the LAST active task serving it, and is
        never touched again afterwards. Returns True if the unit was sent."""
        serving = [t for t in (active_tasks or self.task_order) if t in u["serving"]]
        if task not in serving or u["block"] in self.skip_blocks:
            return False
        for t, alt in u["alts"]:
            if t == task:
                fold(u["main"], alt)
        if task != serving[-1]:
            return False
        self.reduce_tensor(u["main"], u["block"])
        return True

    def reduce_tensor(self, t: torch.Tensor, block_idx: Optional[int] = None):
        """All-reduce (SUM, async) one complete slice of a bucket -- a backbone row's share of block 0's bucket, or a whole block -- on the
        current stream's order. The trainer calls it the moment the slice is final (Averaging._unit_done)."""
        if not self.enabled or (block_idx is not None and block_idx in self.skip_blocks):
            return
        self.handles.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.reduced_bytes += t.numel() * t.element_size()

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []


class Averaging:
    def __init__(self, device, model, hyp: dict, task_ids: Sequence[str], epochs: int = 100, nb: int = 1000, loss_weights=None,
                 linear_lr=False, use_ema=True, rank=-1, world_size=1, sync_bn=False, task_streams: Optional[bool] = None):
        self.device, self.model, self.hyp, self.task_ids = device, model, hyp, list(task_ids)
        self._gt_dropped = None
        # one HIP stream per task pass (see _run_tasks_on_streams). Also under SyncBatchNorm: the host enqueue order of the per-layer
        # collectives is a deterministic function of the model structure, hence the same on every rank
        if task_streams is None:
            task_streams = os.environ.get("CDET_TASK_STREAMS", "1") != "0"
        self.task_streams = bool(task_streams) and torch.device(device).type == "cuda"
        self.rank, self.world_size = rank, world_size
        model.sync_bn = bool(sync_bn)  # SyncBatchNorm: per-layer statistics all-reduced over the ranks (reference train.py:140-143)
        # round 4: those small all-reduces as peer-write kernels over HIP IPC (peer_exchange.py) when asked for (CDET_SYNCBN_PEER=1) and every
        # rank can map every other rank's exchange buffer (one node); otherwise the process group (RCCL) carries them
        model._peer_xchg = None
        if sync_bn and world_size > 1 and torch.device(device).type == "cuda":
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                from ..peer_exchange import try_setup

                model._peer_xchg = try_setup(device, max(rank, 0), world_size)
        self.epochs, self.nb = epochs, nb
        self.nw = max(round(get_hyperparameter(hyp, "warmup_epochs") * nb), 1000)  # averaging.py:58
        self.lr0, self.lrf = get_hyperparameter(hyp, "lr0"), get_hyperparameter(hyp, "lrf")
        self.momentum, self.weight_decay = get_hyperparameter(hyp, "momentum"), get_hyperparameter(hyp, "weight_decay")
        self.lf = (lambda x: (1 - x / (epochs - 1)) * (1.0 - self.lrf) + self.lrf) if linear_lr else one_cycle(1, self.lrf, epochs)
        self.loss_weights = dict(zip(self.task_ids, [1.0] * len(self.task_ids))) if loss_weights is None else dict(loss_weights)
        self.gains = {t: dict(box=get_hyperparameter(hyp, "box", i, t), cls=get_hyperparameter(hyp, "cls", i, t),
                              dfl=get_hyperparameter(hyp, "dfl", i, t)) for i, t in enumerate(self.task_ids)}
        self.ema = ModelEMA(model) if (use_ema and rank in (-1, 0)) else None
        self.epoch = 0
        self.steps = 0
        self.lib = L.load()
        # ---- slot table: one slot per trainable parameter (+ EMA-only slots for BN running statistics)
        g0, g1, g2 = get_param_groups(model)
        group_of = {id(p): 0 for p in g0}
        group_of.update({id(p): 1 for p in g1})
        group_of.update({id(p): 2 for p in g2})
        self.serving = {c.index: list(c.serving_tasks.keys()) for c in model.controllers}
        names = dict(model.named_parameters())
        ema_sd = self.ema.ema.state_dict() if self.ema else {}
        self.slots_meta = []
        model._drop_plans()  # plans pre-bind gradient pointers: compile them after the buckets below exist
        buckets: Dict[int, torch.Tensor] = {}
        for bi, block in enumerate(model.blocks):
            ps = list(block.parameters())  # frozen parameters get a (zero) bucket slice too: they may be unfrozen later
            if not ps:
                continue
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=device)
            buckets[bi] = flat
            off = 0
            for p in ps:
                g = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                model._pgrad[id(p)] = g
                p.grad = g
        # Per-task gradient buckets on the blocks several tasks share: every serving task but the first accumulates into a bucket of its own, folded
        # into the block's bucket in task order after the passes (model._merge_alt_grads: g_A + g_B, the sum the shared buffer held before). With
        # them -- and the running-statistics updates of the later task deferred behind each block (engine.Plan.deferred_stats) -- the task passes of
        # an iteration share no read-modify-write state: on two streams they overlap from the first kernel to the last instead of trailing each other
        # through the trunk (profiles/r04_step_timeline.txt: 27 % of the iteration had two kernels in flight with the per-block event chain).
        # CDET_TASK_DECOUPLE=0: the chained form.
        model._pgrad_alt, model._alt_pairs = {}, []
        self._shared_blocks = set()
        if os.environ.get("CDET_TASK_DECOUPLE", "1") != "0" and torch.device(device).type == "cuda":
            for bi, flat in buckets.items():
                for j, t in enumerate(self.serving.get(bi, ())):
                    if j == 0:
                        continue
                    alt = torch.zeros_like(flat)
                    off = 0
                    for p in model.blocks[bi].parameters():
                        model._pgrad_alt[(id(p), t)] = alt[off:off + p.numel()].view_as(p)
                        off += p.numel()
                    model._alt_pairs.append((flat, alt))
                    self._shared_blocks.add(bi)
        # Reduction units: the slices of the buckets that become final at one point of the backward -- every backbone ROW of block 0 (rows 9 -> 0
        # finish in that order; engine.Plan.bwd_sub), every other block as a whole. A unit is folded (per-task buckets -> the block's bucket, task
        # order) and all-reduced the moment the last task serving it has enqueued its gradients, so the trunk's 124 MB travel under the rest of
        # the backward instead of behind it (reference: DDP's bucketed all-reduce overlapped with backward, train.py:182-184).
        self.units: Dict[object, dict] = {}
        alt_of = {}
        for flat, alt in model._alt_pairs:
            alt_of.setdefault(flat.data_ptr(), []).append(alt)
        self._row_alias: Dict[tuple, int] = {}
        for bi, flat in buckets.items():
            serving = list(self.serving.get(bi, ()))
            alts = list(zip(serving[1:], alt_of.get(flat.data_ptr(), [])))
            if bi == 0 and hasattr(model.blocks[0], "model"):
                off, rows = 0, []
                for li, layer in enumerate(model.blocks[0].model):
                    n = sum(p.numel() for p in layer.parameters())
                    if n:
                        rows.append((li, off, n))
                    off += n
                assert off == flat.numel(), "block 0's bucket is not the concatenation of its rows' parameters"
                # the row units are slices of the bucket, folded with cdet_accumulate_clear, which wants 16-byte aligned pointers: true when every
                # row holds a multiple of 4 floats (widths from make_divisible(8)). A model that breaks it keeps block 0 as ONE unit instead of
                # raising from a hook in the middle of a backward
                if all(o % 4 == 0 for _, o, _ in rows):
                    for li, o, n in rows:
                        self.units[(0, li)] = dict(block=0, main=flat[o:o + n], alts=[(t, a[o:o + n]) for t, a in alts], serving=serving)
                else:
                    self.units[0] = dict(block=0, main=flat, alts=alts, serving=serving)
                    self._row_alias[(0, rows[0][0])] = 0  # (the backward finishes the rows last to first: the whole bucket is complete at row 0's mark)
            else:
                self.units[bi] = dict(block=bi, main=flat, alts=alts, serving=serving)
        self._unit_events: Dict[tuple, "torch.cuda.Event"] = {}
        self._unit_count: Dict[object, int] = {}
        self._in_streams = False
        self._fold_stream = None
        self.trace_cb = None
        for k, p in names.items():
            bi = _block_of(k)
            self.slots_meta.append(dict(p=p, g=model._pgrad[id(p)], mom=torch.zeros_like(p), ema=ema_sd.get(k), group=group_of[id(p)],
                                        div=max(len(self.serving[bi]), 1), key=k, stepped=False))
        if self.ema:
            msd = dict(model.named_buffers())
            for k, v in msd.items():
                if v.dtype.is_floating_point and k in ema_sd:
                    self.slots_meta.append(dict(p=v, g=None, mom=None, ema=ema_sd[k], group=-1, div=1, key=k))
        # The optimizer's TAIL: the slots of the blocks no other task shares (necks + heads: ~70 % of the parameters) sit behind the shared trunk's in the
        # table, so that the update can run as two launches -- the trunk's on the current stream, the rest on a side stream under the NEXT
        # iteration's trunk kernels (train_step(defer_tail=True)); every pass waits for the tail in front of its first block outside the trunk.
        self._early_blocks = frozenset(self._shared_blocks) if os.environ.get("CDET_LATE_PACK", "1") != "0" else frozenset()
        self.slots_meta.sort(key=lambda m: 0 if _block_of(m["key"]) in self._early_blocks else 1)  # (stable: block order inside each half)
        self.n_head_slots = sum(1 for m in self.slots_meta if _block_of(m["key"]) in self._early_blocks)
        self._tail_stream = None
        self._tail_event = None
        self._tail_pending = False
        self.n_slots = len(self.slots_meta)
        self._slots_host = (L.ParamSlot * self.n_slots)()
        self._slots_dev = torch.empty(C.sizeof(self._slots_host), dtype=torch.uint8, device=device)
        self._norm_buf = torch.zeros(1 + 32 * self.n_slots, dtype=torch.float32, device=device)
        # GradScaler (reference trainers/averaging.py:61: amp.GradScaler(enabled=cuda), driven at 158 / 207 / 219-220), device-resident so that no
        # step needs a host sync: {scale, growth_tracker, skipped steps, found_inf of the last step}. fp16 plans (model.half(): the reference's own
        # AMP dtype) scale the loss -- init 65536, x2 every 2000 good steps, /2 on a skipped step (torch's defaults); bf16 plans have fp32's
        # exponent range and keep scale 1 (growth off). The found-inf SKIP applies to both: a non-finite gradient anywhere leaves weights and
        # momentum untouched, zeroes the gradients, still runs the EMA lerp (csrc/optim.hip).
        self.loss_scaling = model.compute_dtype == torch.float16
        self._scaler = torch.tensor([65536.0 if self.loss_scaling else 1.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.scaler_growth, self.scaler_backoff, self.scaler_interval = 2.0, 0.5, (2000 if self.loss_scaling else 0)
        self._slot_key = None
        self.reducer = GradReducer(buckets, self.serving, self.task_ids)
        self.group_sizes = [len(g2), len(g0), len(g1)]  # optimizer.param_groups order of the reference: bias, decay, bn

    # ---------------------------------------------------------------------------------------------------- schedule
    def lrs(self, ni: int, epoch: int):
        """Per-group learning rates [g0, g1, g2] and momentum incl. warm-up (reference base_trainer.py:100-112, with its
        quirk: the optimizer's group index 2 -- the BN-weight group -- is the one treated as "bias")."""
        base = self.lr0 * self.lf(epoch)
        if ni <= self.nw:
            xi = [0, self.nw]
            wb = get_hyperparameter(self.hyp, "warmup_bias_lr")
            by_opt_index = [float(np.interp(ni, xi, [wb if j == 2 else 0.0, base])) for j in range(3)]
            mom = float(np.interp(ni, xi, [get_hyperparameter(self.hyp, "warmup_momentum"), self.momentum]))
            return [by_opt_index[1], by_opt_index[2], by_opt_index[0]], mom  # opt index 0 = g2, 1 = g0, 2 = g1
        return [base, base, base], self.momentum

    # ---------------------------------------------------------------------------------------------------- step pieces
    def forward_backward(self, task: str, batch: dict, n_max: Optional[int] = None, active_tasks=None):
        self.join_tail()
        g = self._pass_steps(task, batch, n_max, active_tasks, None)
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value  # (every unit this pass completed has been folded into p.grad and sent by its hook: _unit_done)

    def _unit_done(self, key, task: str):
        """Hook behind the backward launches of a reduction unit (a backbone row, or a whole block) of `task`'s pass, on that pass's stream.
        Sequential passes: fold this task's own bucket of the unit into the block's bucket right here (task order = stream order: g_A, + g_B,
        + g_C -- the sum one shared buffer would hold), and all-reduce the unit when this was the last active task serving it; a unit is never
        touched again once it is sent. Task streams: the unit is complete when the LAST serving task's stream has enqueued it (host order is
        deterministic, hence the same on every rank) -- then, on a side stream behind all serving tasks' events, the per-task buckets are
        folded in task order and the slice is all-reduced there, while both backward passes run on."""
        key = self._row_alias.get(key, key)
        u = self.units.get(key)
        if self.trace_cb is not None:
            self.trace_cb("unit", key, task)  # (bench.py --dry-comm: where in the host enqueue order the backward passes stand)
        if u is None:
            return
        active = self._active or self.task_ids
        serving = [t for t in active if t in u["serving"]]
        if task not in serving:
            return
        bi = u["block"]
        if bi in self.reducer.skip_blocks:
            return
        if not (self._in_streams and u["alts"]):
            self.reducer.unit_done(u, task, active, lambda main, alt: L.check(self.lib.cdet_accumulate_clear(
                main.data_ptr(), alt.data_ptr(), alt.numel(), torch.cuda.current_stream().cuda_stream), "cdet_accumulate_clear"))
            return
        ev = self._unit_events.get((key, task))
        if ev is None:
            ev = self._unit_events[(key, task)] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        n = self._unit_count[key] = self._unit_count.get(key, 0) + 1
        if n < len(serving):
            return
        self._unit_count[key] = 0
        fs = self._fold_stream
        with torch.cuda.stream(fs):
            for t in serving:
                fs.wait_event(self._unit_events[(key, t)])
            for t, alt in u["alts"]:
                if t in serving:
                    L.check(self.lib.cdet_accumulate_clear(u["main"].data_ptr(), alt.data_ptr(), alt.numel(), fs.cuda_stream), "cdet_accumulate_clear")
            self.reducer.reduce_tensor(u["main"], bi)

    def _pass_steps(self, task: str, batch: dict, n_max, active_tasks, fired):
        """One task pass: fused forward + criterion + backward; gradients accumulate. Returns loss items (device tensor[5])."""
        img = batch["img"]
        if self.model.compute_dtype == torch.float32:
            # a full_precision() model (cerberusdet_amd/precise.py): the same pass at the reference's own precision -- forward, the fused criterion on its
            # fp32 head maps, the full-precision backward into the same .grad buckets; every reduction unit the task serves is then complete at once
            # (no overlap with the backward: this path exists for parity, tests/test_gpu_full_precision.py)
            return self._pass_full_precision(task, batch, n_max, active_tasks)
        plan = self.model.get_plan(task, img.shape, img.dtype, training=True)
        if not plan.hooks:
            for idx in {i for i, _ in plan.bwd_groups}:
                for key in ([(idx, row) for row, _ in plan.bwd_sub[idx]] if idx in plan.bwd_sub else [idx]):
                    plan.hooks[key] = (lambda k, t=task: self._unit_done(k, t))
        self._active = active_tasks
        yield from plan.iter_forward(img, fired)
        if self._gt_dropped is None:
            self._gt_dropped = torch.zeros(1, dtype=torch.int32, device=img.device)
        gt = pad_targets(batch, img.shape[0], (img.shape[2], img.shape[3]), img.device, n_max=n_max, dropped=self._gt_dropped)
        loss5 = plan.loss(task, gt, self.gains[task], grad_scale=float(self.loss_weights[task]), scaler=self._scaler if self.loss_scaling else None)
        yield from plan.iter_backward(fired)
        return loss5

    def _pass_full_precision(self, task: str, batch: dict, n_max, active_tasks):
        from ..ops import det_loss

        img = batch["img"]
        model = self.model
        plan = model.full_precision_plan([task], img, training=True)
        self._active = active_tasks
        with torch.no_grad():
            plan.run(img)
            if self._gt_dropped is None:
                self._gt_dropped = torch.zeros(1, dtype=torch.int32, device=img.device)
            gt = pad_targets(batch, img.shape[0], (img.shape[2], img.shape[3]), img.device, n_max=n_max, dropped=self._gt_dropped)
            head = model.get_head(task)
            loss5, dfe, _ = det_loss(plan.feats[task], gt, head.nc, self.gains[task], [float(s) for s in head.stride],
                                     grad_scale=float(self.loss_weights[task]), grad_dtype=torch.float32)
            for d, g in zip(plan.dfeats[task], dfe):
                d.copy_(g)
            plan.run_backward()
        for key, u in self.units.items():
            if task in u["serving"]:
                self._unit_done(key, task)
        return loss5

    def _run_tasks_on_streams(self, active, batches, n_max, out):
        """The task passes of one iteration on one HIP stream each. The passes are independent except on the blocks they share
        (read-modify-write of the shared BatchNorm running statistics in forward and of the shared gradient buffers in backward):
        there a per-block event chain keeps the reference's task order, so the results are bit-identical to the sequential
        schedule. What the overlap buys: one pass's latency-bound launches (BN partial-sum kernels, ~400 per pass) and the
        partly filled last round of its convolution grids run under the other pass's kernels (measured 113.8 -> 101.9 ms)."""
        from ..engine import BlockSync, task_stream

        cur = torch.cuda.current_stream()
        key = tuple(active)
        cache = self.__dict__.setdefault("_stream_sync", {})
        if key not in cache:
            streams = [task_stream(self.device, k, len(active)) for k in range(len(active))]  # the process-wide lane streams (engine.py)
            syncs = {t: BlockSync() for t in active}
            for idx, ts in self.serving.items():
                chain = [t for t in active if t in ts]
                for a_, b_ in zip(chain[:-1], chain[1:]):
                    ef, eb = torch.cuda.Event(), torch.cuda.Event()
                    syncs[a_].done_fwd[idx], syncs[b_].wait_fwd[idx] = ef, ef
                    syncs[a_].done_bwd[idx], syncs[b_].wait_bwd[idx] = eb, eb
            cache[key] = (streams, syncs)
        streams, syncs = cache[key]
        plans = []
        for t in active:
            img = batches[t]["img"]
            plan = self.model.get_plan(t, img.shape, img.dtype, training=True)
            # the shared trunk's operands are packed here, on the current stream, before the fork (by the first plan that finds them stale);
            # every pass packs its own branch itself in front of its first block outside the trunk, on its OWN stream, behind the optimizer's
            # tail (engine.Plan.iter_forward) -- under the trunk kernels of the iteration
            plan.early_blocks = self._early_blocks
            plan.tail_event = self._tail_event if self._tail_pending else None
            if self._early_blocks:
                plan.refresh_weights("early")
            else:
                plan.refresh_weights()
            plan.attach_grads()
            plans.append(plan)
        # Block-interleaved enqueue: every round lets each task enqueue its next block on its own stream. The GPU-side order is set by
        # the streams and events alone (results are bit-identical to the sequential schedule); the HOST order matters for the
        # collectives of SyncBatchNorm and the gradient reducer, which one communicator serialises in enqueue order -- enqueueing task A's
        # whole pass first would hold task B's first all-reduce behind A's last. A task whose next block waits for an event that has
        # not been recorded in this iteration yet (a shared block the earlier task has not reached) is skipped for the round.
        fired: set = set()
        from ..engine import lane_stream
        if self._fold_stream is None:
            self._fold_stream = lane_stream(self.device, len(self.task_ids) + 1)  # (the lane behind the task streams; eval plans use it as a side lane, never at the same time)
        self._unit_count = {}
        self._in_streams = True
        try:
            gens = []
            for t, st, plan in zip(active, streams, plans):
                plan.block_sync = syncs[t]
                st.wait_stream(cur)
                gens.append((t, st, self._pass_steps(t, batches[t], n_max, active, fired)))
            while gens:
                progressed = False
                for item in list(gens):
                    t, st, g = item
                    with torch.cuda.stream(st):
                        try:
                            req = next(g)
                        except StopIteration as e:
                            out[t] = e.value
                            gens.remove(item)
                            progressed = True
                            continue
                    progressed = progressed or req is None
                if not progressed:
                    raise RuntimeError("task schedule cannot make progress: a block waits for an event no active task records")
        finally:
            self._in_streams = False
            for plan in plans:
                plan.block_sync = None
        for st in streams:
            cur.wait_stream(st)
        cur.wait_stream(self._fold_stream)  # the folds of the shared units (their all-reduces are joined by reducer.wait())

    def set_shared_frozen(self, frozen: bool):
        """--freeze-shared-till-epoch (reference trainers/averaging.py:100-103, models/cerberus.py:885-925): the blocks that serve
        every task stop training -- their parameters take part in neither the clipping norm nor the update (a frozen slot is an
        EMA-only slot); their BatchNorms keep normalising with BATCH statistics (train-form forward, the reference leaves model.train()
        on, trainers/averaging.py:106) while their running statistics stay untouched, and the engine compiles them without a backward."""
        from ..models import CerberusDet

        (CerberusDet.freeze_shared_layers if frozen else CerberusDet.unfreeze_shared_layers)(self.model)
        self._slot_key = None
        self.reducer.skip_blocks = {i for i, b in enumerate(self.model.blocks)
                                    if any(True for _ in b.parameters()) and not any(p.requires_grad for p in b.parameters())}

    def join_tail(self):
        """Order the current stream behind the optimizer's deferred tail (train_step(defer_tail=True)). Call before anything reads weights, momentum
        or EMA of the blocks outside the shared trunk on the current stream: validation, checkpoints, state_dict(), tests."""
        if self._tail_pending:
            torch.cuda.current_stream().wait_event(self._tail_event)
            self._tail_pending = False

    def optimizer_step(self, lrs, momentum, n_serving: Optional[Dict[int, int]] = None, idle_blocks=(), defer_tail: bool = False):
        """idle_blocks: blocks none of whose serving tasks ran this iteration (--skip-batches). The reference's zero_grad() leaves
        their gradients None (torch >= 2.0 default set_to_none), so torch's SGD skips them entirely -- no weight decay, no momentum
        coasting, no momentum-buffer initialisation; here their slots become EMA-only for this step and are not marked stepped."""
        self.join_tail()  # (a previous deferred tail still owns its slots)
        with comm_span(self.model, "grad_wait"):
            self.reducer.wait()
        idle = set(idle_blocks)
        live = [m["g"] is not None and m["p"].requires_grad and _block_of(m["key"]) not in idle for m in self.slots_meta]
        fresh = sum(1 for m, a in zip(self.slots_meta, live) if a and not m.get("stepped", True))
        # (the learning rates are launch arguments, not table entries: the table is rebuilt only when its structure changes)
        key = (fresh, tuple(sorted(n_serving.items())) if n_serving else None, hash(tuple(live)))
        if key != self._slot_key:
            for i, m in enumerate(self.slots_meta):
                s = self._slots_host[i]
                s.p, s.n = m["p"].data_ptr(), m["p"].numel()
                s.g = m["g"].data_ptr() if live[i] else None  # frozen parameter: EMA-only slot
                s.mom = m["mom"].data_ptr() if m["mom"] is not None else None
                s.ema = m["ema"].data_ptr() if m["ema"] is not None else None
                if m["group"] >= 0:
                    s.group = m["group"]
                    s.weight_decay = self.weight_decay if m["group"] == 0 else 0.0
                    div = m["div"] if n_serving is None else max(n_serving.get(_block_of(m["key"]), 1), 1)
                    s.inv_div = 1.0 / div
                s.first_step = int(not m.get("stepped", True))  # momentum buffer starts as the first clipped gradient (torch SGD)
            self._slots_dev.copy_(torch.frombuffer(bytearray(bytes(self._slots_host)), dtype=torch.uint8), non_blocking=False)
            self._slot_key = key
        st = torch.cuda.current_stream().cuda_stream
        sc = self._scaler.data_ptr() if self.loss_scaling else None  # (unscale only when the loss was scaled; the skip needs the norm alone)
        L.check(self.lib.cdet_grad_sqnorm(self._slots_dev.data_ptr(), self.n_slots, self._norm_buf.data_ptr(), sc, st), "cdet_grad_sqnorm")
        n_head = self.n_head_slots if (defer_tail and 0 < self.n_head_slots < self.n_slots) else self.n_slots
        if n_head < self.n_slots:
            from ..engine import lane_stream
            if self._tail_stream is None:
                self._tail_stream = lane_stream(self.device, len(self.task_ids) + 2)
                self._tail_event = torch.cuda.Event()
                self._norm_event = torch.cuda.Event()
            self._norm_event.record(torch.cuda.current_stream())  # the tail forks HERE: behind the reduced gradients and the clipping norm
        d = 0.0
        if self.ema:
            self.ema.updates += 1
            d = self.ema.decay(self.ema.updates)
        lr_arr = (C.c_float * len(lrs))(*[float(v) for v in lrs])
        # (without loss scaling the skip bookkeeping -- skipped-step counter, found_inf flag = words 2, 3 of the state -- rides on the update launch itself)
        cnt = None if self.loss_scaling else self._scaler.data_ptr() + 8
        L.check(self.lib.cdet_sgd_ema_step(self._slots_dev.data_ptr(), n_head, self._norm_buf.data_ptr(), 10.0, lr_arr, len(lrs),
                                           float(momentum), float(d), sc, cnt, st), "cdet_sgd_ema_step")
        if n_head < self.n_slots:
            # the tail: the same kernel over the slots of the unshared blocks, on a side stream behind the clipping norm -- it runs under the next
            # iteration's trunk (HBM-bound update beside MFMA-bound convolutions); same arithmetic per slot, same results
            ts = self._tail_stream
            ts.wait_event(self._norm_event)
            L.check(self.lib.cdet_sgd_ema_step(self._slots_dev.data_ptr() + n_head * C.sizeof(L.ParamSlot), self.n_slots - n_head,
                                               self._norm_buf.data_ptr(), 10.0, lr_arr, len(lrs), float(momentum), float(d), sc, None, ts.cuda_stream),
                    "cdet_sgd_ema_step")
            if self.loss_scaling:  # scaler.update() behind BOTH update launches (the tail reads the old scale too)
                L.check(self.lib.cdet_scaler_update(self._scaler.data_ptr(), self._norm_buf.data_ptr(), self.scaler_growth, self.scaler_backoff,
                                                    self.scaler_interval, ts.cuda_stream), "cdet_scaler_update")
            self._tail_event.record(ts)
            self._tail_pending = True
        elif self.loss_scaling:
            L.check(self.lib.cdet_scaler_update(self._scaler.data_ptr(), self._norm_buf.data_ptr(), self.scaler_growth, self.scaler_backoff,
                                                self.scaler_interval, st), "cdet_scaler_update")
        if fresh:
            for m, a in zip(self.slots_meta, live):
                if a:
                    m["stepped"] = True
        self.model.mark_weights_changed()
        self.steps += 1

    def train_step(self, batches: Dict[str, dict], ni: Optional[int] = None, n_max: Optional[int] = None, defer_tail: bool = False):
        """batches: {task: {"img": [N,3,H,W] uint8|float on device, "batch_idx", "cls", "bboxes"}}; tasks missing from the dict
        are skipped this iteration (the reference's --skip-batches). Returns {task: loss items tensor[5]}."""
        ni = self.steps if ni is None else ni
        lrs, mom = self.lrs(ni, self.epoch)
        active = [t for t in self.task_ids if t in batches]
        out = {}
        if n_max is None and active:
            # the reference sizes the padded targets per task pass (`counts.max()`, utils/loss.py:117 -- a host sync inside every pass);
            # here ONE sync before the task streams fork sizes them for all tasks (padding rows are masked: results do not depend on it)
            counts = [torch.bincount(batches[t]["batch_idx"].reshape(-1).long(), minlength=batches[t]["img"].shape[0]).max()
                      for t in active if batches[t]["batch_idx"].numel()]
            n_max = max(int(torch.stack(counts).max()), 1) if counts else 1
        if self.task_streams and len(active) > 1 and self.model.compute_dtype != torch.float32:
            self._run_tasks_on_streams(active, batches, n_max, out)
        else:
            self.join_tail()
            for t in active:
                out[t] = self.forward_backward(t, batches[t], n_max=n_max, active_tasks=active)
        n_serving, idle = None, ()
        if len(active) != len(self.task_ids):
            n_serving = {i: max(len([t for t in ts if t in active]), 1) for i, ts in self.serving.items()}
            idle = [i for i, ts in self.serving.items() if ts and not any(t in active for t in ts)]
        # defer_tail (training loops; the caller joins with join_tail() before it reads weights outside a train_step): the update of the unshared
        # blocks overlaps the next iteration's trunk. Only with the task streams -- the sequential schedule packs everything up front.
        defer = bool(defer_tail) and self.task_streams and len(self.task_ids) > 1 and len(active) == len(self.task_ids)
        self.optimizer_step(lrs, mom, n_serving, idle, defer_tail=defer)
        if self._tail_pending and not defer:
            self.join_tail()
        return out

    def scaler_state(self):
        """GradScaler state (host sync): dict(scale, growth_tracker, skipped_steps, found_inf) -- `skipped_steps` counts the optimizer steps a
        non-finite gradient has cancelled so far (reference: scaler.step() returns without stepping, trainers/averaging.py:219)."""
        self.join_tail()
        v = self._scaler.tolist()
        return dict(scale=v[0], growth_tracker=int(v[1]), skipped_steps=int(v[2]), found_inf=bool(v[3]))

    def check_targets(self):
        """Call where the training loop synchronises anyway (loss read-back): raises if a label did not fit the `n_max` given to
        train_step() -- the sync-free target padding drops such labels (and counts them) instead of overwriting others."""
        if self._gt_dropped is not None:
            n = int(self._gt_dropped.item())
            if n:
                self._gt_dropped.zero_()
                raise RuntimeError(f"{n} label(s) exceeded n_max in train_step(): pass a larger n_max (or None to size it per batch)")
        px = getattr(self.model, "_peer_xchg", None)
        if px is not None:
            px.check()  # a SyncBatchNorm exchange that timed out waiting for a peer

    # ---------------------------------------------------------------------------------------------------- resume
    def state_dict(self):
        self.join_tail()
        return self._state_dict()

    def _state_dict(self):
        """Everything `--resume` needs besides the model weights (reference utils/models_manager.py:262-308 saves optimizer, EMA +
        updates, epoch, best fitness): momentum buffers and their first-step flags per parameter, EMA weights and update count,
        iteration counters."""
        return dict(format="cerberusdet_amd/trainer/1", epoch=self.epoch, steps=self.steps,
                    momentum={m["key"]: m["mom"].detach().cpu() for m in self.slots_meta if m.get("mom") is not None},
                    stepped={m["key"]: bool(m.get("stepped", True)) for m in self.slots_meta if m.get("mom") is not None},
                    ema=({k: v.detach().cpu() for k, v in self.ema.ema.state_dict().items()} if self.ema else None),
                    ema_updates=(self.ema.updates if self.ema else 0), scaler=self._scaler.detach().cpu())

    def load_state_dict(self, sd):
        assert sd.get("format") == "cerberusdet_amd/trainer/1", "not a cerberusdet_amd trainer state"
        self.epoch, self.steps = int(sd["epoch"]), int(sd["steps"])
        with torch.no_grad():
            for m in self.slots_meta:
                if m.get("mom") is not None and m["key"] in sd["momentum"]:
                    m["mom"].copy_(sd["momentum"][m["key"]])
                    m["stepped"] = bool(sd["stepped"][m["key"]])
            if self.ema and sd.get("ema") is not None:
                self.ema.ema.load_state_dict(sd["ema"])
                self.ema.updates = int(sd["ema_updates"])
            if sd.get("scaler") is not None:
                self._scaler.copy_(sd["scaler"])
        self._slot_key = None
