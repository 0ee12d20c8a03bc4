"""N > 1 path on CPU (gloo, world_size 2): the block-bucketed gradient reduction schedule of the trainer.

The kernels themselves need the GPU; what is checked here is the distributed algebra and schedule:
  * SUM all-reduce of locally accumulated gradients == the reference's DDP(avg) * world_size per task pass
    (trainers/averaging.py:162-163), because reduction is linear;
  * a block's bucket is reduced exactly once per iteration, after the LAST task that serves it (shared blocks after task 2,
    task-1 branch blocks already after task 1), and only blocks on an executed path are reduced;
  * the synthetic-batch sharding of bench.py gives every rank distinct data.
"""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cerberusdet_amd.trainers.averaging import GradReducer

    tasks = ["a", "b"]
    serving = {0: ["a", "b"], 1: ["a", "b"], 2: ["a"], 3: ["b"], 4: []}
    buckets = {i: torch.zeros(8) for i in range(5)}
    red = GradReducer(buckets, serving, tasks)
    assert red.enabled
    log = []
    # local per-task gradient contributions g[task][block]; rank-dependent
    g = {t: {i: torch.full((8,), float((rank + 1) * (10 if t == "a" else 1) * (i + 1))) for i in range(5)} for t in tasks}
    for t in tasks:
        order = [i for i in (3, 2, 1, 0) if t in serving[i]]  # backward visits blocks in reverse
        for i in order:
            buckets[i] += g[t][i]
            n0 = len(red.handles)
            red.on_block_backward(i, t, tasks)
            if len(red.handles) > n0:
                log.append((t, i))
    red.wait()
    want = {i: sum((r + 1) * (10 if t == "a" else 1) * (i + 1) for r in range(world) for t in tasks if t in serving[i]) for i in range(5)}
    ok = all(torch.allclose(buckets[i], torch.full((8,), float(want[i]))) for i in range(4)) and float(buckets[4].abs().sum()) == 0.0
    # skip-batches iteration: only task "b" active -> shared blocks reduce after b, a-only blocks are not touched
    for b in buckets.values():
        b.zero_()
    log2 = []
    for i in (3, 1, 0):
        buckets[i] += g["b"][i]
        n0 = len(red.handles)
        red.on_block_backward(i, "b", ["b"])
        if len(red.handles) > n0:
            log2.append(("b", i))
    red.wait()
    ok2 = bool(torch.allclose(buckets[0], torch.full((8,), float(sum((r + 1) * 1 for r in range(world)))))) and float(buckets[2].abs().sum()) == 0.0
    q.put((rank, bool(ok), log, bool(ok2), log2, red.reduced_bytes))
    dist.destroy_process_group()


def test_block_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, ok, log, ok2, log2, nbytes in res:
        assert ok and ok2, (rank, log, log2)
        # task a: its own branch (2) reduces immediately, shared blocks wait; task b: its branch then the shared blocks
        assert log == [("a", 2), ("b", 3), ("b", 1), ("b", 0)], log
        assert log2 == [("b", 3), ("b", 1), ("b", 0)], log2
        assert nbytes == 7 * 8 * 4


def _worker_deferred(rank, world, port, q):
    """The decoupled schedule (trainers/averaging.py, round 4): on the blocks several tasks serve, every task but the first accumulates into a bucket
    of its own; the block's bucket is complete only after the fold, so the reducer must NOT send it from the block hook but from reduce_deferred."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cerberusdet_amd.trainers.averaging import GradReducer

    tasks = ["a", "b"]
    serving = {0: ["a", "b"], 1: ["a", "b"], 2: ["a"], 3: ["b"]}
    buckets = {i: torch.zeros(8) for i in range(4)}
    alt = {i: torch.zeros(8) for i in (0, 1)}          # task b's buckets on the shared blocks
    red = GradReducer(buckets, serving, tasks)
    red.deferred = {0, 1}
    g = {t: {i: torch.full((8,), float((rank + 1) * (10 if t == "a" else 1) * (i + 1))) for i in range(4)} for t in tasks}
    log = []
    for t in tasks:
        for i in [i for i in (3, 2, 1, 0) if t in serving[i]]:
            (alt[i] if (t == "b" and i in alt) else buckets[i]).add_(g[t][i])
            n0 = len(red.handles)
            red.on_block_backward(i, t, tasks)
            if len(red.handles) > n0:
                log.append((t, i))
    red.wait()
    exclusive_ok = all(torch.allclose(buckets[i], torch.full((8,), float(sum((r + 1) * (10 if i == 2 else 1) * (i + 1) for r in range(world))))) for i in (2, 3))
    shared_untouched = all(torch.allclose(buckets[i], g["a"][i]) for i in (0, 1))  # still local: nothing was sent before the fold
    for i in (0, 1):                                    # the fold (model._merge_alt_grads), then the deferred reduction
        buckets[i] += alt[i]
        alt[i].zero_()
    n0 = len(red.handles)
    red.reduce_deferred(tasks)
    sent = len(red.handles) - n0
    red.wait()
    want = {i: sum((r + 1) * 11 * (i + 1) for r in range(world)) for i in (0, 1)}
    shared_ok = all(torch.allclose(buckets[i], torch.full((8,), float(want[i]))) for i in (0, 1))
    # forward_backward()'s form: only the blocks THIS task completes (only_last_task); task "a" completes none of the shared ones
    n0 = len(red.handles)
    red.reduce_deferred(tasks, only_last_task="a")
    none_for_a = len(red.handles) == n0
    q.put((rank, log, bool(exclusive_ok), bool(shared_untouched), sent, bool(shared_ok), bool(none_for_a)))
    dist.destroy_process_group()


def test_deferred_buckets_of_the_decoupled_schedule_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_deferred, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, log, exclusive_ok, shared_untouched, sent, shared_ok, none_for_a in res:
        assert log == [("a", 2), ("b", 3)], log            # the block hook sends the exclusive blocks only
        assert exclusive_ok and shared_untouched and sent == 2 and shared_ok and none_for_a, (rank, sent)


def test_bench_sharding_gives_distinct_rank_data():
    import bench

    a = bench.synth_batch(0, 0, 0, 2, 20, 64, "cpu")
    b = bench.synth_batch(1, 0, 0, 2, 20, 64, "cpu")
    c = bench.synth_batch(0, 1, 0, 2, 19, 64, "cpu")
    assert not torch.equal(a["img"], b["img"]) and not torch.equal(a["img"], c["img"])
    assert a["img"].dtype == torch.uint8 and a["bboxes"].shape == (16, 4) and float(a["bboxes"][:, :2].min()) >= 0.2
    assert torch.equal(a["img"], bench.synth_batch(0, 0, 0, 2, 20, 64, "cpu")["img"])  # deterministic
