"""Training entry point with the reference's flags and process model (ai-forever/CerberusDet cerberusdet/train.py:279-419):
one process per GPU (`python -m torch.distributed.run --nproc-per-node N cerberusdet_amd/train.py ...`), env LOCAL_RANK /
RANK / WORLD_SIZE, `train(hyp, opt, device, train_dataset=None, val_dataset=None)`, `run(**kwargs)`.

Scope: the hot path -- model, loss, backward, gradient all-reduce, optimizer, and the validation arithmetic
(val.run: NMS at the reference's val settings, device matcher, AP) -- is native here. Plotting and MLflow / TensorBoard logging
are out of scope; `train_dataset` / `val_dataset` are per-task iterables that yield the reference's batch dicts
({"img": uint8 [N,3,H,W], "cls", "bboxes" (xywh in [0,1]), "batch_idx"}): `--data synthetic` provides the benchmark's generator,
`--data <yaml>` the YOLO-txt loader of cerberusdet_amd.data with the reference's training augmentation (mosaic / affine / mixup /
HSV / flips) rendered by one GPU kernel per batch (`--no-augment`: letterbox only). That loader decodes images with PIL on the
training thread (`--workers` is accepted and ignored): it exists for parity of the data path, its throughput is host-bound and is
not a measured result -- the benchmark numbers are synthetic batches resident in HBM.

Per epoch (reference trainers/base_trainer.py:114-194, utils/models_manager.py:262-294): rank 0 validates every task on the EMA
weights (unless --noval, always on the final epoch), fitness = 0.1 * mAP@.5 + 0.9 * mAP@.5:.95 (utils/metrics.py:28-34) per task and
averaged over the tasks; `weights/{task}_best.pt` on a new per-task best, `weights/best.pt` on a new mean best, `weights/last.pt` always;
--patience stops early like the reference's EarlyStopping (utils/torch_utils.py:257-279).
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # task-pass streams + RCCL streams need more than the default 4 hardware queues

import torch  # noqa: E402
import torch.distributed as dist
import yaml

FILE = Path(__file__).resolve()
ROOT = FILE.parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from cerberusdet_amd.utils.torch_utils import parse_device, select_device  # noqa: E402

LOCAL_RANK = int(os.getenv("LOCAL_RANK", -1))
RANK = int(os.getenv("RANK", -1))
WORLD_SIZE = int(os.getenv("WORLD_SIZE", 1))


def parse_opt(known=False):
    p = argparse.ArgumentParser()
    p.add_argument("--weights", type=str, default="", help="initial cerberusdet_amd checkpoint")
    p.add_argument("--cfg", type=str, default=str(ROOT / "cerberusdet_amd/models/cfg/v8x_2task.yaml"), help="model.yaml path")
    p.add_argument("--data", type=str, default="synthetic", help="dataset.yaml path or 'synthetic'")
    p.add_argument("--hyp", type=str, default="", help="hyperparameters path")
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--batch-size", type=str, default="32", help="batch size PER GPU, one value or 'a,b,c' per task")
    p.add_argument("--imgsz", "--img", "--img-size", type=int, default=640)
    p.add_argument("--resume", nargs="?", const=True, default=False)
    p.add_argument("--nosave", action="store_true")
    p.add_argument("--noval", action="store_true", help="only validate the final epoch")
    p.add_argument("--use-multi-labels", action="store_true", help="Loading multiple labels for boxes, if available")
    p.add_argument("--use-soft-labels", action="store_true", help="Class probability based on annotation votes")
    p.add_argument("--labels-from-xml", action="store_true", help="Load labels from xml files")
    p.add_argument("--no-augment", action="store_true", help="--data <yaml>: plain letterboxed images instead of the reference's training augmentation "
                   "(mosaic, random affine, mixup, HSV, flips with the hyper-parameters of --hyp; the reference always augments its training "
                   "loaders, utils/train_utils.py:22-29), which is rendered on the GPU (cerberusdet_amd/augment.py)")
    p.add_argument("--device", default="")
    p.add_argument("--sync-bn", action="store_true", help="SyncBatchNorm: per-layer statistics all-reduced over the ranks")
    p.add_argument("--workers", type=int, default=8, help="accepted for CLI compatibility: batches come from the caller's iterables / the synthetic generator")
    p.add_argument("--project", default="runs/train")
    p.add_argument("--name", default="exp")
    p.add_argument("--exist-ok", action="store_true")
    p.add_argument("--linear-lr", action="store_true")
    p.add_argument("--patience", type=int, default=30, help="EarlyStopping patience (epochs without improvement of the mean fitness); 0 disables it")
    p.add_argument("--local_rank", "--local-rank", type=int, default=-1)
    p.add_argument("--single-cls", action="store_true")
    p.add_argument("--freeze-shared-till-epoch", type=int, default=0)
    p.add_argument("--skip-batches", action="store_true")
    p.add_argument("--iters-per-epoch", type=int, default=20, help="synthetic data: iterations per epoch")
    p.add_argument("--val-iters", type=int, default=2, help="synthetic data: validation batches per task and epoch")
    p.add_argument("--tasks", type=str, default="voc,objects365_animals")
    p.add_argument("--nc", type=str, default="20,19")
    return p.parse_known_args()[0] if known else p.parse_args()


DEFAULT_HYP = dict(lr0=0.00309, lrf=0.0956, momentum=0.952, weight_decay=0.00037, warmup_epochs=2.04, warmup_momentum=0.898,
                   warmup_bias_lr=0.0502, box=7.5, cls=0.5, dfl=1.5)


def fill_tasks_parameters(model, hyp, imgsz, names=None):
    """hyp scaling + attribute contract of the reference (utils/models_manager.py:122-153): box *= 3/nl, cls *= (imgsz/640)^2*3/nl."""
    nl = 3
    hyp = dict(hyp)
    for k, f in (("box", 3.0 / nl), ("cls", (imgsz / 640) ** 2 * 3.0 / nl)):
        hyp[k] = [v * f for v in hyp[k]] if isinstance(hyp[k], list) else hyp[k] * f
    model.hyp = hyp
    model.nc = {t: model.get_head(t).nc for t in model.heads}
    model.names = names or {t: [str(i) for i in range(model.get_head(t).nc)] for t in model.heads}
    return hyp


def fitness(results) -> float:
    """Weighted combination of [P, R, mAP@.5, mAP@.5:.95] (reference utils/metrics.py:28-34)."""
    return 0.0 * results[0] + 0.0 * results[1] + 0.1 * results[2] + 0.9 * results[3]


class EarlyStopping:
    """Reference utils/torch_utils.py:257-279."""

    def __init__(self, patience=30):
        self.best_fitness, self.best_epoch = 0.0, 0
        self.patience = patience or float("inf")
        self.possible_stop = False

    def __call__(self, epoch, fit):
        if fit >= self.best_fitness:  # >= to allow for the early zero-fitness stage of training
            self.best_epoch, self.best_fitness = epoch, fit
        delta = epoch - self.best_epoch
        self.possible_stop = delta >= (self.patience - 1)
        return delta >= self.patience


def reduce_precision_like_reference(model):
    """`model.half().float()` of the reference (train.py:160, rank 0, not when resuming): every floating-point parameter and buffer is
    rounded to fp16 once before the ranks are synchronised; the EMA copy made earlier keeps the un-rounded values. (CerberusDet.half()
    here switches the compute dtype instead, so the rounding is spelled out.)"""
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if t.is_floating_point():
                t.copy_(t.half().float())
    model.mark_weights_changed()


def train(hyp, opt, device, train_dataset=None, val_dataset=None):
    from cerberusdet_amd import val as validate
    from cerberusdet_amd.models import CerberusDet
    from cerberusdet_amd.trainers import Averaging

    tasks = opt.tasks.split(",")
    data_nc = [int(v) for v in opt.nc.split(",")]  # classes in the label files
    nc = [1] * len(data_nc) if opt.single_cls else data_nc  # --single-cls (utils/models_manager.py:84-87): one class "item" per task
    bs = [int(v) for v in str(opt.batch_size).split(",")]
    bs = bs * len(tasks) if len(bs) == 1 else bs
    cfg = yaml.safe_load(open(opt.cfg))
    torch.manual_seed(0)
    model = CerberusDet(tasks, nc, cfg=cfg, verbose=RANK in (-1, 0))
    if cfg.get("cerber"):
        model.sequential_split(cfg["cerber"], "cpu")
    if opt.weights:
        ck = torch.load(opt.weights, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["state_dict"], strict=False)
    model = model.to(device).train()
    names = None
    if train_dataset is None:
        if opt.data == "synthetic":
            import bench

            def gen(ti, t, n=None, seed0=0):
                i = 0
                while n is None or i < n:
                    yield bench.synth_batch(max(RANK, 0), ti, seed0 + i % 4, bs[ti], nc[ti], opt.imgsz, device)
                    i += 1
            train_dataset = {t: gen(ti, t) for ti, t in enumerate(tasks)}
            if val_dataset is None:
                val_dataset = {t: (lambda ti=ti, t=t: gen(ti, t, opt.val_iters, 100)) for ti, t in enumerate(tasks)}
            nb = opt.iters_per_epoch
        else:
            from cerberusdet_amd import data as cdata

            train_dataset, val_dataset_y, names = cdata.datasets_from_yaml(opt.data, tasks, data_nc, bs, opt.imgsz, rank=max(RANK, 0), world_size=WORLD_SIZE,
                                                                           single_cls=opt.single_cls,
                                                                           augment=not getattr(opt, "no_augment", False), hyp=hyp,
                                                                           labels_from_xml=getattr(opt, "labels_from_xml", False),
                                                                           use_multi_labels=getattr(opt, "use_multi_labels", False),
                                                                           use_soft_labels=getattr(opt, "use_soft_labels", False))
            val_dataset = val_dataset if val_dataset is not None else val_dataset_y
            nb = max(len(d) for d in train_dataset.values())
    else:
        nb = max(len(d) if hasattr(d, "__len__") else opt.iters_per_epoch for d in train_dataset.values())
    hyp = fill_tasks_parameters(model, hyp, opt.imgsz, names)
    # the EMA inside the trainer copies the model BEFORE the fp16 rounding below, like the reference (utils/models_manager.py:231)
    trainer = Averaging(device, model, hyp, tasks, epochs=max(opt.epochs, 2), nb=nb, linear_lr=opt.linear_lr, rank=RANK, world_size=WORLD_SIZE,
                        sync_bn=opt.sync_bn and WORLD_SIZE > 1)
    if RANK in (-1, 0) and not opt.resume:
        reduce_precision_like_reference(model)
    if WORLD_SIZE > 1:
        for t in list(model.state_dict().values()):  # what DDP's constructor does in the reference (train.py:182-184)
            dist.broadcast(t, src=0)
        model.mark_weights_changed()
    iters = {t: iter(d) for t, d in train_dataset.items()}
    # --skip-batches (reference trainers/averaging.py:50-54,144-146): a task with a shorter dataset is visited every
    # max_len // len iterations only; the optimizer step then divides by the number of tasks that really ran
    iters_per_task = [1] * len(tasks)
    if opt.skip_batches:
        lens = [len(d) if hasattr(d, "__len__") else nb for d in train_dataset.values()]
        iters_per_task = [max(max(lens) // max(n, 1), 1) for n in lens]
        if RANK in (-1, 0):
            print(f"viewing tasks iteration frequency: {iters_per_task}")
    out_dir = Path(opt.project) / opt.name
    wdir = out_dir / "weights"
    start_epoch = 0
    best_fitness, best_fitness_per_task = 0.0, {t: 0.0 for t in tasks}
    stopper = EarlyStopping(opt.patience)
    if opt.resume:  # reference train.py:359-372 + utils/models_manager.py:296-308: weights, optimizer, EMA, epoch
        ck_path = Path(opt.resume) if isinstance(opt.resume, str) else wdir / "last.pt"
        ck = torch.load(str(ck_path), map_location="cpu", weights_only=False)
        if "trainer" not in ck:
            raise ValueError(f"{ck_path} holds weights only: it cannot resume a run (use --weights)")
        model.load_state_dict(ck["model_state_dict"])
        model.mark_weights_changed()
        trainer.load_state_dict(ck["trainer"])
        start_epoch = trainer.epoch + 1
        best_fitness = float(ck.get("best_fitness", 0.0))
        best_fitness_per_task.update(ck.get("best_fitness_per_task", {}))
        stopper.best_fitness, stopper.best_epoch = best_fitness, int(ck.get("best_epoch", trainer.epoch))
        if RANK in (-1, 0):
            print(f"resuming {ck_path} at epoch {start_epoch} (iteration {trainer.steps})")
    results, items = {}, {}
    val_results = {}
    for epoch in range(start_epoch, opt.epochs):
        trainer.epoch = epoch
        if epoch < opt.freeze_shared_till_epoch:  # reference trainers/averaging.py:100-103
            trainer.set_shared_frozen(True)
        elif 0 < opt.freeze_shared_till_epoch == epoch:
            trainer.set_shared_frozen(False)
        t0 = time.time()
        for i in range(nb):
            batches = {}
            for ti, t in enumerate(tasks):
                if i % iters_per_task[ti] != 0:
                    continue
                try:
                    batches[t] = next(iters[t])
                except StopIteration:
                    iters[t] = iter(train_dataset[t])
                    batches[t] = next(iters[t])
                batches[t] = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batches[t].items()}
            items.update(trainer.train_step(batches, ni=i + nb * epoch, defer_tail=os.environ.get("CDET_DEFER_TAIL", "0") == "1"))
        trainer.join_tail()  # (no-op unless the optimizer's tail was deferred: measured slower, off by default -- trainers/averaging.py)
        torch.cuda.synchronize()
        trainer.check_targets()
        results = {t: [float(v) for v in items[t].tolist()] for t in tasks if t in items}
        if RANK in (-1, 0):
            ips = nb * sum(bs) * WORLD_SIZE / (time.time() - t0)
            print(f"epoch {epoch}: " + "  ".join(f"{t}: box {r[0]:.4f} cls {r[1]:.4f} dfl {r[2]:.4f}" for t, r in results.items()) + f"  [{ips:.1f} img/s]")
        # ---- validation, fitness, checkpoints: rank 0 only, like the reference (train.py:218-226, base_trainer.py:114-194)
        final_epoch = epoch + 1 == opt.epochs or stopper.possible_stop
        stop = False
        if RANK in (-1, 0):
            fit_per_task = {}
            if val_dataset is not None and (not opt.noval or final_epoch):
                eval_model = trainer.ema.ema if trainer.ema else model
                if trainer.ema:  # ema.update_attr(model, include=[... "nc", "hyp", "names", "stride" ...]) of the reference
                    for a in ("nc", "hyp", "names", "stride"):
                        if hasattr(model, a):
                            setattr(eval_model, a, getattr(model, a))
                for t in tasks:
                    vd = val_dataset[t]
                    r = validate.run(eval_model, t, vd() if callable(vd) else vd, single_cls=opt.single_cls)
                    val_results[t] = (r["mp"], r["mr"], r["map50"], r["map"])
                    fit_per_task[t] = fitness(val_results[t])
                    print(f"epoch {epoch} val {t}: P {r['mp']:.4f} R {r['mr']:.4f} mAP@.5 {r['map50']:.4f} mAP@.5:.95 {r['map']:.4f} ({r['seen']} images)")
                    if fit_per_task[t] > best_fitness_per_task[t]:
                        best_fitness_per_task[t] = fit_per_task[t]
                        if not opt.nosave or final_epoch:  # tracks the task's best model (not meant for resuming, base_trainer.py:158-170)
                            save_training_checkpoint(wdir / f"{t}_best.pt", model, trainer, epoch, best_fitness, best_fitness_per_task, stopper.best_epoch)
                last_fitness = sum(fit_per_task.values()) / len(fit_per_task)
                best_fitness = max(best_fitness, last_fitness)
                is_best = best_fitness == last_fitness  # base_trainer.py:177-183: ties (fitness still 0) also write best.pt
                stop = stopper(epoch, last_fitness)
            else:
                is_best = False
            if not opt.nosave or final_epoch:
                save_training_checkpoint(wdir / "last.pt", model, trainer, epoch, best_fitness, best_fitness_per_task, stopper.best_epoch)
                if is_best:
                    save_training_checkpoint(wdir / "best.pt", model, trainer, epoch, best_fitness, best_fitness_per_task, stopper.best_epoch)
        if WORLD_SIZE > 1:  # every rank leaves the loop together (reference train.py:251-257 broadcasts the stop flag)
            flag = torch.tensor([int(stop)], device=device)
            dist.broadcast(flag, src=0)
            stop = bool(flag.item())
        if stop:
            if RANK in (-1, 0):
                print(f"Stopping training early: no improvement in the last {opt.patience} epochs (best epoch {stopper.best_epoch}, weights/best.pt)")
            break
    if val_results:
        results = {t: val_results.get(t, ()) + tuple(results.get(t, ())) for t in tasks}
    return results, epoch if opt.epochs > start_epoch else opt.epochs - 1


def save_training_checkpoint(path, model, trainer, epoch=-1, best_fitness=0.0, best_fitness_per_task=None, best_epoch=0):
    """The inference checkpoint of cerberusdet_inference.save_checkpoint (EMA weights when there is an EMA, like the reference's
    models_manager.py:262-290) plus what --resume needs: the raw model weights, the trainer state, epoch and best fitness. Built once
    and written through a temporary file + os.replace, so an interrupted save never leaves a truncated checkpoint behind."""
    from cerberusdet_amd.cerberusdet_inference import checkpoint_dict

    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    ck = checkpoint_dict(trainer.ema.ema if trainer.ema else model, getattr(model, "names", None))
    ck["state_dict"] = {k: v.detach().cpu() for k, v in ck["state_dict"].items()}
    ck["model_state_dict"] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ck["trainer"] = trainer.state_dict()
    ck.update(epoch=epoch, best_fitness=float(best_fitness), best_fitness_per_task=dict(best_fitness_per_task or {}), best_epoch=int(best_epoch))
    tmp = path.with_suffix(path.suffix + ".tmp")
    torch.save(ck, str(tmp))
    os.replace(str(tmp), str(path))


def main(opt):
    # reference train.py:375-388: under a launcher the rank's GPU is LOCAL_RANK, otherwise `select_device(opt.device)`; --device cpu raises here
    # (no CPU path) instead of silently training on a GPU
    if LOCAL_RANK != -1:
        parse_device(opt.device)  # (still refuses 'cpu' / malformed values)
        assert torch.cuda.device_count() > LOCAL_RANK, "insufficient GPUs for the DDP command"
        device = torch.device("cuda", LOCAL_RANK)
    else:
        device = select_device(opt.device)
    torch.cuda.set_device(device)
    if LOCAL_RANK != -1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl" if dist.is_nccl_available() else "gloo", device_id=device)
    hyp = dict(DEFAULT_HYP)
    if opt.hyp:
        hyp.update(yaml.safe_load(open(opt.hyp)))
    res = train(hyp, opt, device)
    if WORLD_SIZE > 1 and RANK == 0:
        print("Destroying process group... ")
    if LOCAL_RANK != -1:
        dist.destroy_process_group()
    return res


def run(**kwargs):
    opt = parse_opt(True)
    for k, v in kwargs.items():
        setattr(opt, k, v)
    return main(opt)


if __name__ == "__main__":
    main(parse_opt())
