"""The training entry point end to end on the MI355X (reference cerberusdet/train.py:42-277, trainers/base_trainer.py:114-194):
`train.run(...)` on the small 2-task YOLOv8n config -- synthetic batches and a YOLO-txt dataset written to disk -- with per-epoch
validation on the EMA weights, fitness / best.pt / {task}_best.pt / last.pt, --resume, and the fp16 rounding of the initial weights."""
import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def _opts(tmp_path, **kw):
    from cerberusdet_amd import train as T

    base = dict(cfg=str(T.ROOT / "cerberusdet_amd/models/cfg/v8n_2task.yaml"), epochs=2, batch_size="4", imgsz=128, iters_per_epoch=3, val_iters=1,
                project=str(tmp_path), name="exp", tasks="voc,objects365_animals", nc="20,19", patience=0)
    base.update(kw)
    return base


def test_synthetic_run_validates_saves_and_resumes(tmp_path):
    from cerberusdet_amd import train as T
    from cerberusdet_amd.cerberusdet_inference import attempt_load

    res, epoch = T.run(**_opts(tmp_path))
    assert epoch == 1 and set(res) == {"voc", "objects365_animals"}
    for t, r in res.items():
        assert len(r) >= 4 + 4 and all(np.isfinite(v) for v in r)  # (P, R, mAP@.5, mAP@.5:.95) + the last loss items
        assert 0.0 <= r[2] <= 1.0 and 0.0 <= r[3] <= r[2] + 1e-9
    w = tmp_path / "exp" / "weights"
    assert (w / "last.pt").exists() and not list(w.glob("*.tmp"))
    ck = torch.load(str(w / "last.pt"), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and "trainer" in ck and "best_fitness" in ck and set(ck["best_fitness_per_task"]) == {"voc", "objects365_animals"}
    m = attempt_load(str(w / "last.pt"))  # the inference side of the checkpoint (EMA weights)
    assert set(m.heads) == {"voc", "objects365_animals"}
    # --resume continues at the next epoch with the saved optimizer / EMA state
    res2, epoch2 = T.run(**_opts(tmp_path, epochs=3, resume=True))
    assert epoch2 == 2
    ck2 = torch.load(str(w / "last.pt"), map_location="cpu", weights_only=False)
    assert ck2["epoch"] == 2 and ck2["trainer"]["steps"] == 9


def test_initial_weights_are_rounded_to_fp16_like_the_reference_and_ema_is_not(tmp_path):
    """reference train.py:160 `model.half().float()` after the EMA copy (utils/models_manager.py:231)."""
    from cerberusdet_amd import train as T
    from cerberusdet_amd.models import CerberusDet

    cfg = yaml.safe_load(open(T.ROOT / "cerberusdet_amd/models/cfg/v8n_2task.yaml"))
    torch.manual_seed(0)
    m = CerberusDet(["a", "b"], [3, 4], cfg=cfg, verbose=False).cuda()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    T.reduce_precision_like_reference(m)
    changed = 0
    for k, v in m.state_dict().items():
        if v.is_floating_point():
            assert torch.equal(v, before[k].half().float()), k
            changed += int(not torch.equal(v, before[k]))
        else:
            assert torch.equal(v, before[k])
    assert changed > 50


def test_yolo_txt_dataset_trains_and_validates(tmp_path):
    from PIL import Image

    from cerberusdet_amd import train as T

    rng = np.random.RandomState(0)
    root = tmp_path / "ds"
    for ti, (task, nc) in enumerate((("voc", 20), ("objects365_animals", 19))):
        for split, n in (("train", 6), ("val", 3)):
            (root / task / "images" / split).mkdir(parents=True)
            (root / task / "labels" / split).mkdir(parents=True)
            for i in range(n):
                h, w = [(96, 128), (128, 128), (120, 80)][i % 3]
                img = rng.randint(0, 255, (h, w, 3), dtype=np.uint8)
                rows = []
                for _ in range(2):
                    cx, cy, bw, bh = rng.uniform(0.3, 0.7), rng.uniform(0.3, 0.7), rng.uniform(0.2, 0.4), rng.uniform(0.2, 0.4)
                    c = rng.randint(0, nc)
                    x1, y1, x2, y2 = (int(v) for v in ((cx - bw / 2) * w, (cy - bh / 2) * h, (cx + bw / 2) * w, (cy + bh / 2) * h))
                    img[y1:y2, x1:x2] = (10 * c) % 255  # something class-dependent to look at
                    rows.append(f"{c} {cx:.6f} {cy:.6f} {bw:.6f} {bh:.6f}")
                Image.fromarray(img).save(root / task / "images" / split / f"{i}.png")
                if i != 2:  # one background image without a label file
                    (root / task / "labels" / split / f"{i}.txt").write_text("\n".join(rows) + "\n")
    data = dict(train=[f"voc/images/train", "objects365_animals/images/train"], val=["voc/images/val", "objects365_animals/images/val"],
                nc=[20, 19], names=[[f"v{i}" for i in range(20)], [f"a{i}" for i in range(19)]], task_ids=["voc", "objects365_animals"])
    yaml.safe_dump(data, open(root / "data.yaml", "w"))
    # the loader itself: batch dict contract of the reference (data/datasets.py:440-459)
    from cerberusdet_amd.data import datasets_from_yaml

    tr, va, names = datasets_from_yaml(str(root / "data.yaml"), data["task_ids"], [20, 19], [4, 4], 128)
    assert len(tr["voc"]) == 2 and names["voc"][3] == "v3"
    b = next(iter(va["voc"]))
    # validation loaders are rectangular like the reference's (rect=True, pad=0.5): sorted by aspect ratio, one frame per batch --
    # here the three images span ratios below and above 1, so the frame is the square of ceil(128 / 32 + 0.5) * 32 = 160
    assert b["img"].shape == (3, 3, 160, 160) and b["img"].dtype == torch.uint8 and b["img"].is_cuda
    assert b["bboxes"].shape[1] == 4 and b["cls"].shape[1] == 1 and b["prob"].shape[1] == 1 and b["batch_idx"].tolist() == [0, 0, 1, 1]
    assert float(b["bboxes"].min()) >= 0 and float(b["bboxes"].max()) <= 1 and len(b["ori_shape"]) == 3
    assert b["ori_shape"] == ((96, 128), (128, 128), (120, 80)) and b["ratio_pad"][1][1] == (16.0, 16.0)
    assert int(b["img"][1, :, 0, 0].float().mean()) == 114 and int(b["img"][1, 0, 16, 16]) != 114  # 128x128 sits at (16, 16) of its 160x160 frame
    tr_plain = datasets_from_yaml(str(root / "data.yaml"), data["task_ids"], [20, 19], [4, 4], 128, augment=False)[0]
    bt = next(iter(tr_plain["voc"]))
    assert bt["img"].shape[1:] == (3, 128, 128)  # training frames stay square
    res, epoch = T.run(**_opts(tmp_path, data=str(root / "data.yaml"), name="ds", no_augment=True))
    assert epoch == 1 and (tmp_path / "ds" / "weights" / "last.pt").exists()
    assert all(np.isfinite(v) for r in res.values() for v in r)
    # the same with the reference's training augmentation (mosaic / affine / mixup / HSV / flips rendered on the GPU)
    res, epoch = T.run(**_opts(tmp_path, data=str(root / "data.yaml"), name="ds_aug"))  # the default, as in the reference
    assert epoch == 1 and all(np.isfinite(v) for r in res.values() for v in r)
    res, epoch = T.run(**_opts(tmp_path, data=str(root / "data.yaml"), name="ds_single", single_cls=True, epochs=1))  # --single-cls: nc = 1 heads
    assert epoch == 0 and (tmp_path / "ds_single" / "weights" / "last.pt").exists() and (tmp_path / "ds_aug" / "weights" / "last.pt").exists()
    assert all(np.isfinite(v) for r in res.values() for v in r)
