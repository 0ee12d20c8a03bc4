"""Host side of the --data path (cerberusdet_amd/data.py): file discovery, label parsing and the label geometry of the letterboxed frame.
Expected values were produced by the reference's own xywhn2xyxy / xyxy2xywhn (utils/general.py) for load_image + letterbox(auto=False)
(data/datasets.py:395-407) in the build container; they are constants here because the reference does not travel."""
import numpy as np
import pytest

from cerberusdet_amd.data import img2label_path, letterbox_labels, list_images, read_labels, rect_batch_shapes


def test_label_geometry_matches_reference_known_answers():
    lb = np.array([[0.02, 0.5, 0.1, 0.2], [0.5, 0.5, 0.4, 0.3], [0.9, 0.1, 0.3, 0.3]], np.float32)
    got, shp, geo = letterbox_labels(lb, (375, 500), 640)
    assert geo == (640, 480, 80, 0) and shp == ((375, 500), ((480 / 375, 640 / 500), (0.0, 80.0)))
    want = np.array([[0.035, 0.5, 0.07, 0.15], [0.5, 0.5, 0.4, 0.225], [0.87499925, 0.2, 0.2499985, 0.225]], np.float32)
    assert np.abs(got - want).max() < 1e-6
    # narrow image: the reference truncates the resized size (int(97 * 256 / 333) = 74) before letterboxing
    got, shp, geo = letterbox_labels(lb[1:2], (333, 97), 256)
    assert geo == (74, 256, 0, 91) and shp[1][1] == (91.0, 0.0)
    assert np.abs(got - np.array([[0.5, 0.5, 0.4 * 74 / 256, 0.3]], np.float32)).max() < 1e-6
    empty, _, _ = letterbox_labels(np.zeros((0, 4), np.float32), (480, 640), 128)
    assert empty.shape == (0, 4)


def test_discovery_and_label_parsing(tmp_path):
    (tmp_path / "images" / "a").mkdir(parents=True)
    (tmp_path / "labels" / "a").mkdir(parents=True)
    for n in ("1.jpg", "2.png", "notes.txt"):
        (tmp_path / "images" / "a" / n).write_bytes(b"x")
    files = list_images(str(tmp_path / "images"))
    assert [f.rsplit("/", 1)[1] for f in files] == ["1.jpg", "2.png"]
    lp = img2label_path(files[0])
    assert lp == str(tmp_path / "labels" / "a" / "1.txt") and img2label_path(files[0], use_xml=True).endswith("/labels/a/1.xml")
    open(lp, "w").write("3 0.5 0.5 0.2 0.2\n1 0.25 0.25 0.1 0.1\n")
    lb = read_labels(lp, 20)
    assert lb.shape == (2, 6) and lb[0].tolist() == pytest.approx([3, 1.0, 0.5, 0.5, 0.2, 0.2])
    assert read_labels(img2label_path(files[1]), 20).shape == (0, 6)  # background image
    with pytest.raises(AssertionError):
        read_labels(lp, 3)  # class id beyond nc
    with pytest.raises(FileNotFoundError):
        list_images(str(tmp_path / "labels" / "a" / "none"))


def test_label_files_like_the_reference_verifier(tmp_path):
    """tests/golden/labels.json = the real reference's verify_image_label (data/datasets.py:621-690) on the same label texts: accepted
    rows (incl. the sorted order np.unique leaves when duplicates were removed) or rejection (the reference then drops the image)."""
    import json
    from pathlib import Path

    g = json.load(open(Path(__file__).parent / "golden" / "labels.json"))
    for name, text in g["txt"].items():
        f = tmp_path / f"{name}.txt"
        f.write_text(text)
        want = g["txt_cases"][name]
        if want is None:
            with pytest.raises(Exception):
                read_labels(str(f), 20)
        else:
            got = read_labels(str(f), 20)
            assert got.dtype == np.float32 and np.array_equal(got, np.array(want, np.float32).reshape(-1, 6)), name
    x = tmp_path / "a.xml"
    x.write_text(g["xml"])
    for multi in (False, True):
        for soft in (False, True):
            got = read_labels(str(x), 3, use_xml=True, classnames=g["names"], as_multi_label=multi, as_soft_label=soft)
            want = np.array(g["xml_cases"][f"multi{int(multi)}_soft{int(soft)}"], np.float32)
            assert np.array_equal(got, want), (multi, soft, got, want)


def test_balanced_sampler_draws_like_the_reference():
    """tests/golden/sampler.json = the real reference's BalancedBatchSampler (data/samplers.py:9-101, np.random seeded) on the same label table:
    identical sequence of image indices for the same generator state, images without labels never drawn."""
    import json
    from pathlib import Path

    from cerberusdet_amd.data import balanced_order

    g = json.load(open(Path(__file__).parent / "golden" / "sampler.json"))
    labels = [np.concatenate((np.array(c, np.float32).reshape(-1, 1), np.ones((len(c), 5), np.float32)), 1) for c in g["table"]]
    empty = {i for i, c in enumerate(g["table"]) if not c}
    for seed, want in enumerate(g["epochs"]):
        got = balanced_order(labels, np.random.RandomState(seed)).tolist()
        assert got == want and len(got) == len(labels) and not (set(got) & empty)
    per_class = np.zeros(8)
    for i in g["epochs"][0]:
        for c in g["table"][i]:
            per_class[c] += 1
    present = per_class[per_class > 0]
    assert present.max() <= 4 * present.min()  # the point of the sampler: rare classes are drawn about as often as frequent ones


def test_rectangular_validation_batches_like_the_reference():
    """tests/golden/rect_val.json = the real reference's LoadImagesAndLabels(rect=True, pad=0.5, augment=False) on PNG files of these sizes
    (the validation loaders, utils/train_utils.py:45-57): sort order, per-batch frames, and per image the letterboxed labels and the
    `shapes` entry val uses to map boxes back."""
    import json
    from pathlib import Path

    g = json.load(open(Path(__file__).parent / "golden" / "rect_val.json"))
    order, frames = rect_batch_shapes(g["sizes"], g["batch"], g["imgsz"], 32, 0.5)
    ar = [h / w for h, w in g["sizes"]]
    # the sort is by aspect ratio with an unstable argsort: images of EQUAL ratio may swap places (they do between numpy builds), nothing else
    assert [ar[i] for i in order] == [ar[i] for i in g["order"]] and sorted(order.tolist()) == sorted(g["order"])
    assert [list(f) for f in frames] == g["batch_shapes"]
    for i, want in zip(g["order"], g["items"]):
        rows = np.array(g["label_rows"][i], np.float32).reshape(-1, 5)
        frame = tuple(want["img_hw"])  # the frame of the batch the reference put this image in
        got, shp, (new_w, new_h, top, left) = letterbox_labels(rows[:, 1:], tuple(g["sizes"][i]), g["imgsz"], frame)
        (h0, w0), ((rh, rw), (dw, dh)) = shp
        assert [[h0, w0], [[rh, rw], [dw, dh]]] == want["shapes"], (i, shp, want["shapes"])
        ref = np.array(want["labels"], np.float32).reshape(-1, 6)
        assert len(ref) == len(rows) and (not len(rows) or np.abs(got - ref[:, 2:]).max() < 2e-6), (i, got, ref)
        assert 0 <= left and left + new_w <= frame[1] and 0 <= top and top + new_h <= frame[0]
