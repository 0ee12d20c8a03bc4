"""SyncBatchNorm statistics exchange by peer writes over HIP IPC (csrc/peer_exchange.hip) -- the host side.

The reference converts every BatchNorm to SyncBatchNorm (train.py:140-143): one small all-reduce per layer and direction, 348-363 per iteration of the
YOLOv8x 2-task model even after stage batching (engine.Plan._sbn_group). As RCCL calls each of them is a host-enqueued launch plus a ring protocol for a few
KB on the dependent chain of its layer. `PeerExchange` replaces them by one single-workgroup kernel each: every rank owns one exchange buffer, mapped by
all ranks of the node through IPC handles exchanged ONCE here (over the existing process group); a collective is then "write my row into every peer's
slot, publish the epoch, wait for everybody's epoch in my own buffer, sum the rows in rank order" -- no RCCL call, no host involvement, bit-identical
sums on every rank. The gradient all-reduce stays on RCCL (hundreds of MB per iteration: bandwidth-bound, what RCCL is for).

Slots are handed out by a bump allocator in the order the plan compiler asks for them -- a deterministic function of the model structure, hence the same on
every rank (bench.py --dry-comm asserts that order). CDET_SYNCBN_PEER=0 keeps the RCCL form (the tested fallback; also taken when the IPC set-up fails on
any rank, e.g. ranks on different nodes)."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L


class PeerExchange:
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
        # One exchange at a time per rank, in host-enqueue order (what a communicator does for its collectives): an exchange kernel WAITS in the
        # kernel for its peers, so two of them in flight on two streams of one rank could each sit in front of the kernel the other rank's
        # counterpart is waiting for (streams can share a hardware queue) -- a cycle. Chained by an event, the exchanges of a rank run in the
        # order the plan enqueued them, which is the same on every rank (bench.py --dry-comm asserts that order).
        self._chain_ev = None
        self._chain_last = None
        self._ext = {}
        self._mine = C.c_void_p()
        self._peers = []
        with torch.cuda.device(self.device):
            L.check(self.lib.cdet_peer_alloc(capacity_bytes, C.byref(self._mine)), "cdet_peer_alloc")
            handle = (C.c_ubyte * 64)()
            L.check(self.lib.cdet_peer_export(self._mine, handle), "cdet_peer_export")
            handles = [None] * world
            dist.all_gather_object(handles, bytes(handle), group=group)
            ptrs = []
            for r, h in enumerate(handles):
                if r == rank:
                    ptrs.append(self._mine.value)
                    continue
                p = C.c_void_p()
                buf = (C.c_ubyte * 64).from_buffer_copy(h)
                L.check(self.lib.cdet_peer_import(buf, C.byref(p)), "cdet_peer_import")
                self._peers.append(p)
                ptrs.append(p.value)
            self.table = torch.tensor(ptrs, dtype=torch.int64).to(self.device)
            self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        # nobody may write into a peer's buffer before that peer has zeroed and published it: one barrier after the hand-shake
        dist.barrier(group=group)

    # ---------------------------------------------------------------------------------------------------------------------------------
    def make_call(self, t: torch.Tensor):
        """Launch-list entry (callable taking the raw stream) that SUM-all-reduces the fp32 vector `t` in place over the ranks."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cuda" and t.device.index == self.device.index
        n = t.numel()
        data_bytes, flag_bytes = 2 * self.world * n * 4, 2 * self.world * 4
        data_off = (self._bump + 255) // 256 * 256
        flag_off = data_off + data_bytes
        self._bump = flag_off + flag_bytes
        if self._bump > self.capacity:
            raise RuntimeError(f"PeerExchange: {self._bump} bytes of slots exceed the exchange buffer ({self.capacity}); raise CDET_PEER_XCHG_MB")
        self.n_calls += 1
        state = {"epoch": 0}
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
        """Raises if an exchange timed out waiting for a peer (host sync)."""
        e = int(self.err.item())
        if e:
            raise RuntimeError(f"PeerExchange: rank {self.rank} timed out waiting for rank {e - 1} (a rank died, or the ranks enqueue different exchange sequences)")

    def close(self):
        for p in self._peers:
            self.lib.cdet_peer_close(p)
        self._peers = []
        if self._mine:
            self.lib.cdet_peer_free(self._mine)
            self._mine = C.c_void_p()


def try_setup(device, rank: int, world: int, group=None):
    """PeerExchange on every rank, or None on every rank (RCCL form) when any rank cannot set it up / CDET_SYNCBN_PEER=0."""
    if world <= 1 or os.environ.get("CDET_SYNCBN_PEER", "1") == "0":
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
    px, ok = None, 1
    try:
        px = PeerExchange(device, rank, world, group, capacity_bytes=int(os.environ.get("CDET_PEER_XCHG_MB", "64")) << 20)
    except Exception as e:  # noqa: BLE001 -- any failure means "use RCCL", decided collectively below
        print(f"[cerberusdet_amd] peer exchange unavailable on rank {rank} ({e}); SyncBatchNorm statistics go over the process group", flush=True)
        ok = 0
    if px is not None:
        # known-answer exchange before anything depends on it: rank r contributes r + 1, every rank must read world (world + 1) / 2.
        # A mapping that opens but does not carry peer stores (or flags that never become visible) shows up HERE, as a fall-back to the
        # process group, instead of as timed-out exchanges inside the first training iteration.
        try:
            probe = torch.full((64,), float(rank + 1), dtype=torch.float32, device=px.device)
            with torch.cuda.device(px.device):
                px.make_call(probe)(torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
            px.n_calls -= 1  # (not an exchange of any plan)
            if int(px.err.item()) != 0 or not bool((probe == world * (world + 1) / 2).all()):
                raise RuntimeError(f"known-answer exchange failed (error word {int(px.err.item())}, got {float(probe[0])})")
        except Exception as e:  # noqa: BLE001
            print(f"[cerberusdet_amd] peer exchange self-test failed on rank {rank} ({e}); SyncBatchNorm statistics go over the process group", flush=True)
            px.err.zero_()
            ok = 0
    flags = [None] * world
    dist.all_gather_object(flags, ok, group=group)
    if not all(flags):
        if px is not None:
            px.close()
        return None
    return px
