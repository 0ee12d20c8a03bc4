"""The augmentation kernel (csrc/augment.hip, cdet_mosaic_augment_batch) against the numpy restatement of the reference's pixel chain
(oracle/augment.py: cv2.resize of load_image, the mosaic paste, cv2.warpAffine / cv2.warpPerspective, mixup, augment_hsv, flips, BGR->RGB CHW -- reference
data/datasets.py:361-438,470-527, data/augmentations.py:43-57,151,205-211). Integer / uint8 work: **bit-exact**. The plans come from
cerberusdet_amd/augment.py with the generator states of tests/golden/augment.json, i.e. the reference's own parameters."""
import random

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _plans(name):
    from cerberusdet_amd import augment as A

    c = synth.AUG_CASES[name]
    sizes, labels = synth.aug_dataset(c["seed"], c["n"], c["s"])
    plans = []
    for k in range(c["samples"]):
        rng, nprng = random.Random(c["seed"] * 1000 + k), np.random.RandomState(c["seed"] * 1000 + k)
        plans.append(A.sample_plan(rng, nprng, (7 * k + 3) % c["n"], range(c["n"]), sizes, labels, c["s"], c["hyp"]))
    return c, sizes, plans


@pytest.mark.parametrize("name", list(synth.AUG_CASES))
def test_rendered_batch_equals_the_numpy_restatement(name):
    from cerberusdet_amd import augment as A
    from oracle import augment as OA

    c, sizes, plans = _plans(name)
    images = synth.aug_images(c["seed"], sizes)
    dev_images = [torch.from_numpy(im).to(DEV) for im in images]
    got = A.render_batch(plans, dev_images, c["s"], torch.device(DEV))
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == (len(plans), 3, c["s"], c["s"]) and got.dtype == np.uint8
    n_mix = n_flip = 0
    for k, p in enumerate(plans):
        mos = [([(t.index, t.hw, t.dst, t.src) for t in m.tiles], m.M, m.canvas) for m in p.mosaics]
        want = OA.render(mos, p.mix_ratio, p.hsv_lut, p.flipud, p.fliplr, images, c["s"], perspective=p.mosaics[0].perspective)
        bad = got[k] != want
        assert not bad.any(), f"{name} sample {k}: {int(bad.sum())} of {bad.size} bytes differ, max |diff| {np.abs(got[k].astype(int) - want.astype(int)).max()}"
        n_mix += p.mix_ratio is not None
        n_flip += p.fliplr
        assert (want != 114).mean() > 0.3  # the sample shows image content, not just the border colour
    assert n_mix > 0 or c["hyp"]["mixup"] == 0
    assert n_flip > 0 or c["hyp"]["fliplr"] == 0


def test_augmenting_dataset_yields_the_reference_batch_dict(tmp_path):
    """`TaskDataset(augment=True)`: decoded files -> one launch per batch -> {"img" uint8 [N,3,s,s], "cls", "prob", "bboxes", "batch_idx"} with the
    labels of the plans (collate_fn, data/datasets.py:440-459)."""
    from PIL import Image

    from cerberusdet_amd.data import TaskDataset

    s, n = 96, 7
    sizes, labels = synth.aug_dataset(5, n, s)
    images = synth.aug_images(5, sizes)
    (tmp_path / "images").mkdir()
    (tmp_path / "labels").mkdir()
    for i, (im, lb) in enumerate(zip(images, labels)):
        Image.fromarray(im[:, :, ::-1]).save(tmp_path / "images" / f"{i:03d}.png")
        with open(tmp_path / "labels" / f"{i:03d}.txt", "w") as f:
            for r in lb:
                f.write(f"{int(r[0])} {r[2]:.6f} {r[3]:.6f} {r[4]:.6f} {r[5]:.6f}\n")
    ds = TaskDataset(str(tmp_path / "images"), s, 4, 20, DEV, shuffle=False, augment=True, seed=11)
    batches = list(ds)
    assert len(batches) == 2 and batches[0]["img"].shape == (4, 3, s, s) and batches[1]["img"].shape == (3, 3, s, s)
    for b in batches:
        assert b["img"].dtype == torch.uint8 and b["img"].is_cuda
        nl = b["cls"].shape[0]
        assert b["bboxes"].shape == (nl, 4) and b["prob"].shape == (nl, 1) and b["batch_idx"].shape == (nl,)
        if nl:
            assert float(b["bboxes"].min()) >= 0 and float(b["bboxes"].max()) <= 1 and int(b["batch_idx"].max()) < b["img"].shape[0]
        assert float((b["img"] != 114).float().mean()) > 0.3
    again = list(TaskDataset(str(tmp_path / "images"), s, 4, 20, DEV, shuffle=False, augment=True, seed=11))
    assert torch.equal(again[0]["img"], batches[0]["img"]) and torch.equal(again[0]["bboxes"], batches[0]["bboxes"])  # seeded: reproducible
    # hyp["perspective"] != 0 travels from the hyper-parameter dict to the kernel's warpPerspective branch (augmentations.py:152-153)
    persp = list(TaskDataset(str(tmp_path / "images"), s, 4, 20, DEV, shuffle=False, augment=True, seed=11, hyp={"perspective": 0.001}))
    assert persp[0]["img"].shape == batches[0]["img"].shape and not torch.equal(persp[0]["img"], batches[0]["img"])
    assert float((persp[0]["img"] != 114).float().mean()) > 0.3
    if persp[0]["bboxes"].numel():
        assert float(persp[0]["bboxes"].min()) >= 0 and float(persp[0]["bboxes"].max()) <= 1
