"""SyncBatchNorm statistics exchange by peer writes over HIP IPC (csrc/peer_exchange.hip) -- the host side.

The reference converts every BatchNorm to SyncBatchNorm (train.py:140-143): one small all-reduce per layer and direction, 348-363 per iteration of the
YOLOv8x 2-task model even after stage batching (engine.Plan._sbn_group). As RCCL calls each of them is a host-enqueued launch plus a ring protocol for a few
KB on the dependent chain of its layer. `PeerExchange` replaces them by one single-workgroup kernel each: every rank owns one exchange buffer, mapped by
all ranks of the node through IPC handles exchanged ONCE here (over the existing process group); a collective is then "write my row into every peer's
slot, publish the epoch, wait for everybody's epoch in my own buffer, sum the rows in rank order" -- no RCCL call, no host involvement, bit-identical
sums on every rank. The gradient all-reduce stays on RCCL (hundreds of MB per iteration: bandwidth-bound, what RCCL is for).

Slots are handed out by a bump allocator in the order the plan compiler asks for them -- a deterministic function of the model structure, hence the same on
every rank (bench.py --dry-comm asserts that order); a slot is keyed by (task set, layer(s), direction), so recompiled plans reuse theirs. OPT-IN:
CDET_SYNCBN_PEER=1 (round 5: until the one-kernel form has run on real peers the default is the process-group form, also taken -- collectively -- when
the staged IPC set-up fails on any rank, e.g. ranks on different nodes)."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L


def _all_ok(ok: bool, world: int, group) -> bool:
    """Collective AND of a per-rank flag. Every set-up stage ends with one, on EVERY rank whatever happened locally, so that a rank whose stage
    failed never leaves its peers inside a collective it will not join (the sequence of collectives is the same on all ranks by construction)."""
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok), group=group)
    return all(flags)


class PeerExchange:
    """Built in stages by `try_setup` (alloc + export, import, publish): each stage catches its own failure and the ranks agree on the outcome
    before the next one starts. A directly constructed object (tests, one process = several virtual ranks) is an empty shell until then."""

    def __init__(self, device, rank: int, world: int, group=None, capacity_bytes: int = 64 << 20):
        self.lib = L.load()
        dev = torch.device(device)
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.rank, self.world, self.group = rank, world, group
        self.capacity = capacity_bytes
        # CDET_PEER_XCHG_HOSTSYNC=1: the two halves of an exchange as separate launches around a host barrier -- for ranks that share one GPU
        # (the driver time-slices the kernels of different processes; an in-kernel wait for a peer that cannot run would only time out)
        self.hostsync = os.environ.get("CDET_PEER_XCHG_HOSTSYNC", "0") == "1"
        self.n_calls = 0       # exchanges compiled into launch lists (collectives per iteration = those on the executed plans)
        self._bump = 256       # bytes handed out (the first 256 stay zero)
        self._slots = {}       # key -> (data offset, flag offset, epoch state): a recompiled plan (frozen / unfrozen trunk, another batch shape)
        #                        reuses the slots of the layers it shares with the plans before it instead of growing the buffer
        # One exchange at a time per rank, in host-enqueue order (what a communicator does for its collectives): an exchange kernel WAITS in the
        # kernel for its peers, so two of them in flight on two streams of one rank could each sit in front of the kernel the other rank's
        # counterpart is waiting for (streams can share a hardware queue) -- a cycle. Chained by an event, the exchanges of a rank run in the
        # order the plan enqueued them, which is the same on every rank (bench.py --dry-comm asserts that order).
        self._chain_ev = None
        self._chain_last = None
        self._ext = {}
        self._mine = C.c_void_p()
        self._peers = []
        self.table = self.err = None

    # ---- set-up stages (each may raise; try_setup turns a failure on ANY rank into a collective fall-back) -------------------------------------
    def stage_alloc_export(self) -> bytes:
        with torch.cuda.device(self.device):
            L.check(self.lib.cdet_peer_alloc(self.capacity, C.byref(self._mine)), "cdet_peer_alloc")
            handle = (C.c_ubyte * 64)()
            L.check(self.lib.cdet_peer_export(self._mine, handle), "cdet_peer_export")
            self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        return bytes(handle)

    def stage_import(self, handles):
        ptrs = []
        with torch.cuda.device(self.device):
            for r, h in enumerate(handles):
                if r == self.rank:
                    ptrs.append(self._mine.value)
                    continue
                p = C.c_void_p()
                buf = (C.c_ubyte * 64).from_buffer_copy(h)
                L.check(self.lib.cdet_peer_import(buf, C.byref(p)), "cdet_peer_import")
                self._peers.append(p)
                ptrs.append(p.value)
            self.table = torch.tensor(ptrs, dtype=torch.int64).to(self.device)

    # ---------------------------------------------------------------------------------------------------------------------------------
    def make_call(self, t: torch.Tensor, key=None):
        """Launch-list entry (callable taking the raw stream) that SUM-all-reduces the fp32 vector `t` in place over the ranks.
        key: identity of the exchange (task set, layer(s), direction) -- the same key gets the same slot and continues its epoch count, so the
        plans a long run compiles (frozen / unfrozen trunk, the odd last batch, multi-scale shapes) do not each take a fresh set of slots. Two
        exchanges that can be in flight at the same time must not share a key (the task set is part of it: engine.Plan._allreduce_call)."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cuda" and t.device.index == self.device.index
        n = t.numel()
        slot = self._slots.get((key, n)) if key is not None else None
        if slot is None:
            data_bytes, flag_bytes = 2 * self.world * n * 4, 2 * self.world * 4
            data_off = (self._bump + 255) // 256 * 256
            flag_off = data_off + data_bytes
            if flag_off + flag_bytes > self.capacity:
                raise RuntimeError(f"PeerExchange: {flag_off + flag_bytes} bytes of slots exceed the exchange buffer ({self.capacity}); raise CDET_PEER_XCHG_MB")
            self._bump = flag_off + flag_bytes
            slot = (data_off, flag_off, {"epoch": 0})
            if key is not None:
                self._slots[(key, n)] = slot
        data_off, flag_off, state = slot
        self.n_calls += 1
        lib, ptr, tab, err, world, rank = self.lib, t.data_ptr(), self.table.data_ptr(), self.err.data_ptr(), self.world, self.rank
        d_off, f_off = data_off // 4, flag_off // 4

        hostsync, group = self.hostsync, self.group

        def call(st, t=t):  # (keeps `t` alive)
            state["epoch"] += 1
            ext = self._ext.get(st)
            if ext is None:
                ext = self._ext[st] = torch.cuda.ExternalStream(st, device=self.device) if st else torch.cuda.default_stream(self.device)
                if self._chain_ev is None:
                    self._chain_ev = torch.cuda.Event()
            if self._chain_last is not None and self._chain_last != st:
                ext.wait_event(self._chain_ev)
            if hostsync:  # ranks sharing ONE GPU (tests): publish, host barrier, collect -- a spinning kernel would wait for a time-sliced peer
                rc = lib.cdet_peer_allreduce(ptr, n, tab, world, rank, d_off, f_off, state["epoch"], err, 1, st)
                torch.cuda.synchronize()
                dist.barrier(group=group)
                rc = rc or lib.cdet_peer_allreduce(ptr, n, tab, world, rank, d_off, f_off, state["epoch"], err, 2, st)
            else:
                rc = lib.cdet_peer_allreduce(ptr, n, tab, world, rank, d_off, f_off, state["epoch"], err, 0, st)
            if rc:
                L.check(rc, "cdet_peer_allreduce")
            self._chain_ev.record(ext)
            self._chain_last = st

        call.__name__ = "peer_allreduce"
        return call

    def check(self):
        """Raises if an exchange timed out waiting for a peer (host sync). The error word is sticky: the kernel that timed out has written NaN
        into its vector (the loss shows it at once) and every later check raises again -- the run is not recoverable."""
        e = int(self.err.item())
        if e:
            raise RuntimeError(f"PeerExchange: rank {self.rank} timed out waiting for rank {e - 1} (a rank died, or the ranks enqueue different exchange sequences); "
                               "the statistics of that exchange were poisoned with NaN")

    def close(self):
        for p in self._peers:
            self.lib.cdet_peer_close(p)
        self._peers = []
        if self._mine:
            self.lib.cdet_peer_free(self._mine)
            self._mine = C.c_void_p()


def try_setup(device, rank: int, world: int, group=None):
    """PeerExchange on every rank, or None on every rank (the process-group form). OPT-IN: CDET_SYNCBN_PEER=1 -- the one-kernel exchange has only
    ever run between virtual ranks / processes sharing one GPU; until it has run on real peers over xGMI the default keeps
    SyncBatchNorm's statistics on the process group (RCCL), which simply waits where this one has to bound its wait.
    Every stage below ends in the same collective on every rank, whatever happened locally: a rank whose allocation, IPC import or self-test
    fails reports it and all ranks fall back together (and release what they had built)."""
    if world <= 1 or os.environ.get("CDET_SYNCBN_PEER", "0") != "1":
        return None
    # Ranks that SHARE a GPU (tests; never a production layout) are time-sliced by the driver, not run side by side: the in-kernel wait of
    # the one-kernel exchange would only time out. Unless the split form is asked for explicitly (CDET_PEER_XCHG_HOSTSYNC=1), such a group
    # keeps the process-group exchange.
    import socket

    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ident = (socket.gethostname(), str(getattr(torch.cuda.get_device_properties(idx), "uuid", idx)), idx)
    idents = [None] * world
    dist.all_gather_object(idents, ident, group=group)
    if len(set(idents)) < world and os.environ.get("CDET_PEER_XCHG_HOSTSYNC", "0") != "1":
        return None

    def note(stage, e):
        print(f"[cerberusdet_amd] peer exchange: {stage} failed on rank {rank} ({e}); SyncBatchNorm statistics go over the process group", flush=True)

    px, handle = None, b""
    # stage 1: allocate + export. The handle gather below doubles as this stage's agreement (an empty handle = "failed here").
    try:
        px = PeerExchange(device, rank, world, group, capacity_bytes=int(os.environ.get("CDET_PEER_XCHG_MB", "64")) << 20)
        handle = px.stage_alloc_export()
    except Exception as e:  # noqa: BLE001 -- any failure means "use the process group", decided collectively
        note("allocation / export", e)
        handle = b""
    handles = [None] * world
    dist.all_gather_object(handles, handle, group=group)
    ok = all(len(h) == 64 for h in handles)
    # stage 2: import the peers' buffers (only when every rank exported one), then agree
    good = ok
    if ok:
        try:
            px.stage_import(handles)
        except Exception as e:  # noqa: BLE001
            note("IPC import", e)
            good = False
    ok = _all_ok(good, world, group)
    # nobody may write into a peer's buffer before that peer has zeroed and published it: the agreement above is that barrier.
    # stage 3: known-answer exchange before anything depends on the mapping: rank r contributes r + 1, every rank must read world (world + 1) / 2.
    # A mapping that opens but does not carry peer stores (or flags that never become visible) shows up HERE, not inside the first iteration.
    good = ok
    if ok:
        # (a SHORT wait budget for this one exchange: a mapping that does not carry peer stores must fail start-up in seconds, not spin a kernel
        #  for the 10 minutes a training exchange may wait for a slow rank -- `peer_spin_ms` switch of the library, csrc/switches.h)
        budget = L.get_switch("peer_spin_ms")
        try:
            L.set_switch("peer_spin_ms", int(os.environ.get("CDET_PEER_SELFTEST_MS", "5000")))
            probe = torch.full((64,), float(rank + 1), dtype=torch.float32, device=px.device)
            with torch.cuda.device(px.device):
                px.make_call(probe)(torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
            px.n_calls -= 1  # (not an exchange of any plan)
            if int(px.err.item()) != 0 or not bool((probe == world * (world + 1) / 2).all()):
                raise RuntimeError(f"known-answer exchange failed (error word {int(px.err.item())}, got {float(probe[0])})")
        except Exception as e:  # noqa: BLE001
            note("self-test", e)
            good = False
        finally:
            L.set_switch("peer_spin_ms", budget)
    ok = _all_ok(good, world, group)
    if not ok:
        if px is not None:
            px.close()  # the 64 MB buffer and whatever peers were already mapped
        return None
    return px
