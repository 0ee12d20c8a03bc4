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


def _units(buckets, serving, alts=None, rows=None):
    """Reduction units as trainers.Averaging builds them: block 0 split into `rows` slices, every other block whole."""
    units = {}
    for i, flat in buckets.items():
        al = list(zip(serving[i][1:], (alts or {}).get(i, [])))
        if i == 0 and rows:
            off = 0
            for r, n in enumerate(rows):
                units[(0, r)] = dict(block=0, main=flat[off:off + n], alts=[(t, a[off:off + n]) for t, a in al], serving=serving[i])
                off += n
        else:
            units[i] = dict(block=i, main=flat, alts=al, serving=serving[i])
    return units


def _fold(main, alt):
    main += alt
    alt.zero_()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cerberusdet_amd.trainers.averaging import GradReducer

    tasks = ["a", "b"]
    serving = {0: ["a", "b"], 1: ["a", "b"], 2: ["a"], 3: ["b"], 4: []}
    buckets = {i: torch.zeros(8) for i in range(5)}
    red = GradReducer(buckets, serving, tasks)
    units = _units(buckets, serving)  # the chained form: one shared gradient buffer per block, no per-task buckets
    assert red.enabled
    log = []
    # local per-task gradient contributions g[task][block]; rank-dependent
    g = {t: {i: torch.full((8,), float((rank + 1) * (10 if t == "a" else 1) * (i + 1))) for i in range(5)} for t in tasks}
    for t in tasks:
        order = [i for i in (3, 2, 1, 0) if t in serving[i]]  # backward visits blocks in reverse
        for i in order:
            buckets[i] += g[t][i]
            if red.unit_done(units[i], t, tasks, _fold):
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
        if red.unit_done(units[i], "b", ["b"], _fold):
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


def _worker_rows(rank, world, port, q):
    """The decoupled schedule with per-row reduction units (trainers/averaging.py, round 5): on the blocks several tasks serve every task but the
    first accumulates into a bucket of its own; block 0's bucket travels as one slice per backbone row, each folded (task order) and all-reduced
    as soon as the last serving task has produced it. Against the round-4 form -- fold everything after the passes, all-reduce whole buckets --
    the results must be BIT-identical (random fp32 gradients), and nothing may be sent before its last contribution is in."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cerberusdet_amd.trainers.averaging import GradReducer

    tasks = ["a", "b", "c"]
    serving = {0: ["a", "b", "c"], 1: ["a", "b"], 2: ["a"], 3: ["b"], 4: ["c"]}
    rows = [5, 16, 3, 40]                                 # block 0 = four backbone rows
    size = {0: sum(rows), 1: 24, 2: 8, 3: 8, 4: 8}
    gen = torch.Generator().manual_seed(7 + rank)
    g = {t: {i: torch.randn(size[i], generator=gen) for i in size if t in serving[i]} for t in tasks}

    def run(per_row, active):
        buckets = {i: torch.zeros(size[i]) for i in size}
        alts = {i: [torch.zeros(size[i]) for _ in serving[i][1:]] for i in size}
        red = GradReducer(buckets, serving, tasks)
        units = _units(buckets, serving, alts, rows if per_row else None)
        log = []
        for t in active:
            for i in [i for i in (4, 3, 2, 1, 0) if t in serving[i]]:
                dst = buckets[i] if t == serving[i][0] else alts[i][serving[i].index(t) - 1]
                keys = [(0, r) for r in reversed(range(len(rows)))] if (i == 0 and per_row) else [i]
                off = size[i]
                for k in keys:                                # the backward finishes block 0's rows from the last to the first
                    n = rows[k[1]] if isinstance(k, tuple) else size[i]
                    off -= n
                    dst[off:off + n] += g[t][i][off:off + n]
                    if per_row:
                        if red.unit_done(units[k], t, active, _fold):
                            log.append((t, k))
        if not per_row:                                       # round 4: fold all per-task buckets after the passes, then send whole buckets
            for i in size:
                for a in alts[i]:
                    _fold(buckets[i], a)
            for i in size:
                if any(t in serving[i] for t in active):
                    red.reduce_tensor(buckets[i], i)
        red.wait()
        assert all(float(a.abs().sum()) == 0.0 for al in alts.values() for a in al)
        return buckets, log, red.reduced_bytes

    out = {}
    for name, active in (("all", tasks), ("bc", ["b", "c"]), ("a", ["a"])):
        b_new, log, n_new = run(True, active)
        b_old, _, n_old = run(False, active)
        out[name] = (all(torch.equal(b_new[i], b_old[i]) for i in size), log, n_new == n_old)
    q.put((rank, out))
    dist.destroy_process_group()


def test_row_units_of_the_decoupled_schedule_world2_bit_identical_to_whole_buckets():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_rows, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, out in res:
        for name, (same, log, same_bytes) in out.items():
            assert same and same_bytes, (rank, name)
        # all three tasks: the exclusive blocks leave with their own task, block 1 after b, block 0's rows one by one (last row first) after c
        assert out["all"][1] == [("a", 2), ("b", 3), ("b", 1), ("c", 4), ("c", (0, 3)), ("c", (0, 2)), ("c", (0, 1)), ("c", (0, 0))], out["all"][1]
        # --skip-batches: b and c active -> block 1 is complete after b alone, block 0 after c
        assert out["bc"][1] == [("b", 3), ("b", 1), ("c", 4), ("c", (0, 3)), ("c", (0, 2)), ("c", (0, 1)), ("c", (0, 0))], out["bc"][1]
        assert out["a"][1] == [("a", 2), ("a", 1), ("a", (0, 3)), ("a", (0, 2)), ("a", (0, 1)), ("a", (0, 0))], out["a"][1]


def test_bench_sharding_gives_distinct_rank_data():
    import bench

    a = bench.synth_batch(0, 0, 0, 2, 20, 64, "cpu")
    b = bench.synth_batch(1, 0, 0, 2, 20, 64, "cpu")
    c = bench.synth_batch(0, 1, 0, 2, 19, 64, "cpu")
    assert not torch.equal(a["img"], b["img"]) and not torch.equal(a["img"], c["img"])
    assert a["img"].dtype == torch.uint8 and a["bboxes"].shape == (16, 4) and float(a["bboxes"][:, :2].min()) >= 0.2
    assert torch.equal(a["img"], bench.synth_batch(0, 0, 0, 2, 20, 64, "cpu")["img"])  # deterministic


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    p = subprocess.run([sys.executable, str(root / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return p.returncode, [json.loads(l) for l in lines if l.startswith("{")], lines, p.stderr


def test_bench_gpus2_launcher_path_with_gloo_stub_ranks():
    """`python bench.py --gpus 2 --steps K --warmup W` WITHOUT a launcher (what a user types; the driver wraps it in torch.distributed.run itself):
    the parent must pick a free port, start torch.distributed.run with two ranks on 127.0.0.1, hand the arguments through unchanged, leave stdout
    to rank 0's ONE JSON line and relay the exit code. CDET_BENCH_STUB=1 replaces the GPU step by a stand-in behind the same barrier /
    MAX-over-ranks protocol, rendezvous over gloo -- this path had never executed anywhere."""
    rc, objs, lines, err = _run_bench({"CDET_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and len(objs) == 1, lines  # exactly one line on stdout: rank 0's
    o = objs[0]
    assert o["stub"] and o["n_gpus"] == 2 and o["steps"] == 3 and o["warmup"] == 1 and o["scaling"] == "weak"
    assert o["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5"]
    assert o["sum_of_ranks"] == 3.0                      # both ranks took part in the stand-in's exchange
    assert 1024 < int(o["master_port"]) < 65536
    assert o["value"] == round(5 * 2 * 2 * 3 / (o["ms_per_step"] * 3 / 1e3), 2) or abs(o["value"] - 5 * 2 * 2 / (o["ms_per_step"] / 1e3)) < 0.01 * o["value"]
    # a rank that dies makes the whole command fail (the launcher's exit code reaches the caller), and nothing lands on stdout
    rc, objs, lines, err = _run_bench({"CDET_BENCH_STUB": "fail"}, "--gpus", "2", "--steps", "2", "--warmup", "0")
    assert rc != 0 and objs == []
    # a --gpus / WORLD_SIZE mismatch under an outer launcher is refused, not silently run at another size
    rc, objs, lines, err = _run_bench({"CDET_BENCH_STUB": "1", "WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert rc != 0 and objs == [] and "--gpus 2 but WORLD_SIZE=4" in err
