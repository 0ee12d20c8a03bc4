"""Host helpers of utils/general.py (reference general.py:122-127, 206-213, 272-357): known answers worked out by hand from the
reference's formulas, plus the oracle's scale_boxes on random boxes."""
import math

import numpy as np
import torch

from cerberusdet_amd.utils import general as g


def test_make_divisible_and_img_size():
    assert g.make_divisible(640, 32) == 640
    assert g.make_divisible(641, 32) == 672
    assert g.make_divisible(100.5, 8) == 104      # ceil(12.5625) * 8
    assert g.make_divisible(1, 32) == 32
    assert g.check_img_size(650, 32.0) == 672


def test_one_cycle_endpoints_and_midpoint():
    f = g.one_cycle(1.0, 0.01, 100)
    assert f(0) == 1.0
    assert abs(f(100) - 0.01) < 1e-12
    assert abs(f(50) - 0.505) < 1e-12
    assert abs(f(25) - (1.0 + (0.01 - 1.0) * (1 - math.cos(math.pi / 4)) / 2)) < 1e-12


def test_xywh2xyxy_torch_and_numpy_keep_extra_columns():
    x = torch.tensor([[10.0, 20.0, 4.0, 6.0, 0.9, 3.0]])
    want = torch.tensor([[8.0, 17.0, 12.0, 23.0, 0.9, 3.0]])
    assert torch.equal(g.xywh2xyxy(x), want)
    assert np.array_equal(g.xywh2xyxy(x.numpy()), want.numpy())
    assert x[0, 0] == 10.0  # input untouched


def test_scale_and_clip_boxes_letterbox_round_trip():
    # 720x1280 frame letterboxed to 640x640: gain 0.5, pad (0, 140)
    b = torch.tensor([[0.0, 140.0, 640.0, 500.0], [-5.0, 100.0, 700.0, 600.0]])
    out = g.scale_boxes((640, 640), b, (720, 1280))
    assert out is b
    assert torch.equal(b, torch.tensor([[0.0, 0.0, 1280.0, 720.0], [0.0, 0.0, 1280.0, 720.0]]))
    n = np.array([[10.0, 150.0, 20.0, 160.0]])
    g.scale_boxes((640, 640), n, (720, 1280), ratio_pad=((0.5, 0.5), (0.0, 140.0)))
    assert np.array_equal(n, np.array([[20.0, 20.0, 40.0, 40.0]]))
    from oracle import nms as onms

    rng = np.random.default_rng(0)
    boxes = rng.uniform(-50, 700, size=(32, 4)).astype(np.float32)
    want = onms.scale_boxes((640, 640), boxes.copy(), (480, 854))
    got = g.scale_boxes((640, 640), torch.from_numpy(boxes.copy()), (480, 854)).numpy()
    assert np.allclose(got, want, rtol=0, atol=1e-4)


def test_box_iou_known_values():
    a = torch.tensor([[0.0, 0.0, 10.0, 10.0]])
    b = torch.tensor([[0.0, 0.0, 10.0, 10.0], [5.0, 5.0, 15.0, 15.0], [20.0, 20.0, 30.0, 30.0]])
    iou = g.box_iou(a, b)
    assert iou.shape == (1, 3)
    assert abs(iou[0, 0].item() - 1.0) < 1e-6 and abs(iou[0, 1].item() - 25.0 / 175.0) < 1e-6 and iou[0, 2].item() == 0.0
