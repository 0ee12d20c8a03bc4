"""CPU-only checks of the host side: structure parity with the reference (block numbering, plans, state-dict schema), the
C-ABI library loads and exports every symbol declared in include/cerberus_hip.h, host helpers."""
import copy
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import synth
from util import GOLDEN, load_golden

ROOT = Path(__file__).resolve().parents[1]
KA = json.load(open(GOLDEN / "graph_known_answers.json"))
CFGS = {
    "yolov8x_voc_obj365.yaml": ("v8x_2task.yaml", ["voc", "objects365_animals"], [20, 19]),
    "yolov8x_voc_obj365_animals_tableware.yaml": ("v8x_3task.yaml", ["voc", "objects365_animals", "objects365_tableware"], [20, 19, 12]),
    "yolov8x.yaml": ("v8x.yaml", ["voc"], [20]),
}


def test_library_exports_every_declared_symbol():
    from cerberusdet_amd import _lib as L

    hdr = (ROOT / "include" / "cerberus_hip.h").read_text()
    declared = set(re.findall(r"\b(cdet_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = L.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/cerberus_hip.h but not exported"
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    assert lib.cdet_version() == 1


def test_no_cpu_fallback():
    """The product path must fail loudly without the GPU instead of silently computing elsewhere."""
    import yaml

    from cerberusdet_amd.models import CerberusDet
    from cerberusdet_amd.utils.general import non_max_suppression

    cfg = yaml.safe_load(open(ROOT / "cerberusdet_amd/models/cfg/v8n_2task.yaml"))
    m = CerberusDet(["a", "b"], [3, 4], cfg=cfg, verbose=False)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m(torch.zeros(1, 3, 64, 64))
        with pytest.raises(RuntimeError):
            non_max_suppression(torch.zeros(1, 8, 10))
    with pytest.raises(RuntimeError):
        m.blocks[0].model[0](torch.zeros(1, 3, 64, 64))  # structural modules have no eager path
    # kept signatures do not silently change meaning: the reference would run device="cpu" on the CPU (cerberusdet_inference.py:30-36)
    # and would fill rep_tensors for retain_tensors / retain_all (cerberus.py:866-872)
    from cerberusdet_amd.cerberusdet_inference import CerberusDetInference

    with pytest.raises(RuntimeError, match="not supported"):
        CerberusDetInference("does-not-matter.pt", device="cpu")
    for kw in (dict(retain_tensors=True), dict(retain_all=True)):
        with pytest.raises(NotImplementedError):
            m(torch.zeros(1, 3, 64, 64), **kw)
    # the full-precision path (cerberusdet_amd/precise.py) has no CPU form either, and the compiled 16-bit plans refuse an fp32 compute dtype
    assert m.float() is m and m.compute_dtype == torch.bfloat16            # .float() stays a no-op: parameters are fp32 masters already
    assert m.full_precision() is m and m.compute_dtype == torch.float32
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="MI355X"):
            m.eval()(torch.zeros(1, 3, 64, 64))
    with pytest.raises(NotImplementedError, match="16 bits"):
        m.get_plan("a", (1, 3, 64, 64), torch.float32, training=False)
    assert m.half().compute_dtype == torch.float16 and m.bfloat16().compute_dtype == torch.bfloat16


@pytest.mark.parametrize("ref_name", list(CFGS))
def test_structure_matches_reference(ref_name):
    import yaml

    from cerberusdet_amd.models import CerberusDet

    ours, tasks, nc = CFGS[ref_name]
    cfg = yaml.safe_load(open(ROOT / "cerberusdet_amd/models/cfg" / ours))
    ka = KA[ref_name]
    m = CerberusDet(tasks, nc, cfg=copy.deepcopy(cfg), verbose=False)
    if cfg.get("cerber"):
        m.sequential_split(cfg["cerber"], "cpu")
    assert len(m.blocks) == ka["n_blocks"] and m.heads == ka["heads"]
    for t in tasks:
        assert m.execution_plan(t)[0] == ka["plans"][t]
    assert m.execution_plan(tasks)[0] == ka["plan_all"]
    assert sorted(m.branching_points) == ka["branching_points"]
    assert [float(s) for s in m.stride] == ka["stride"]
    assert sum(p.numel() for p in m.parameters()) == ka["n_params"]
    assert len(m.state_dict()) == ka["n_state_keys"]
    assert {str(c.index): list(c.serving_tasks.keys()) for c in m.controllers} == ka["serving"]
    assert [type(b).__name__ for b in m.blocks] == ka["block_types"]
    h = m.get_head(tasks[0])
    assert all(abs(float(h.cv3[i][-1].bias[0].detach()) - ka["cls_bias_init"][i]) < 1e-5 for i in range(3))
    # task-filtered parameters(): only blocks on that task's path
    n_task = sum(p.numel() for p in m.parameters(task_ids=tasks[0]))
    assert n_task == sum(p.numel() for i in ka["plans"][tasks[0]] for p in m.blocks[i].parameters())
    copy.deepcopy(m)


@pytest.mark.parametrize("which", ["2task", "3task"])
def test_split_clones_weights_and_rewires(which):
    from cerberusdet_amd.models import CerberusDet

    meta = json.load(open(GOLDEN / ("model_tiny2.json" if which == "2task" else "model_tiny3.json")))
    m = CerberusDet(meta["tasks"], meta["nc"], cfg=copy.deepcopy(meta["cfg"]), verbose=False)
    n0 = len(m.blocks)
    for i, b in enumerate(m.blocks):
        for p in b.parameters():
            p.data.fill_(float(i))
    m.sequential_split(meta["cfg"]["cerber"], "cpu")
    ka = KA["clones"][which]
    assert len(m.blocks) == ka["n_blocks"]
    for i in range(n0, len(m.blocks)):
        ps = list(m.blocks[i].parameters())
        src = int(ps[0].flatten()[0]) if ps else None
        assert src == ka["clone_source"][str(i)]
    for i, b in enumerate(m.blocks):
        if i > 0:
            got = [list(x) if isinstance(x, tuple) else int(x) for x in b.f]
            assert got == ka["block_f"][str(i)], (i, got, ka["block_f"][str(i)])
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == meta["state_shapes"]


def test_pad_targets_has_no_cpu_path():
    from cerberusdet_amd.utils.loss import pad_targets

    b = synth.make_batch(2, 3, 20, 5)
    with pytest.raises(RuntimeError, match="MI355X"):
        pad_targets({k: torch.from_numpy(v) for k, v in b.items()}, 2, (96, 128), "cpu")


def test_lr_schedule_and_param_groups_match_oracle():
    import yaml

    from cerberusdet_amd.models import CerberusDet
    from cerberusdet_amd.trainers.averaging import get_param_groups
    from oracle import optim as oo

    meta = json.load(open(GOLDEN / "trainer.json"))
    mmeta = json.load(open(GOLDEN / "model_tiny2.json"))
    m = CerberusDet(mmeta["tasks"], mmeta["nc"], cfg=copy.deepcopy(mmeta["cfg"]), verbose=False)
    m.sequential_split(mmeta["cfg"]["cerber"], "cpu")
    g0, g1, g2 = get_param_groups(m)
    assert [len(g2), len(g0), len(g1)] == meta["param_group_sizes"]
    names = {id(p): k for k, p in m.named_parameters()}
    for grp, ps in enumerate((g0, g1, g2)):
        assert all(oo.param_group(names[id(p)]) == grp for p in ps)
    # warm-up / schedule (reference base_trainer.py:100-112) -- check the host formula against the oracle's restatement
    from cerberusdet_amd.trainers.averaging import Averaging

    tr = Averaging.__new__(Averaging)
    tr.hyp, tr.nw, tr.lr0, tr.momentum, tr.lf = meta["hyp"], 1000, meta["hyp"]["lr0"], meta["hyp"]["momentum"], (lambda e: 0.9)
    for ni in (0, 1, 500, 1000):
        lrs, mom = tr.lrs(ni, 0)
        wl, wm = oo.warmup_lr(ni, 1000, meta["hyp"]["lr0"], 0.9)
        assert np.allclose(lrs, wl) and abs(mom - wm) < 1e-12
    assert tr.lrs(1001, 0)[0] == [meta["hyp"]["lr0"] * 0.9] * 3



@pytest.mark.parametrize("which", ["tiny2", "tiny3"])
def test_yolo_checkpoint_remap_matches_reference_golden(which):
    """utils/ckpt_utils.py:dict_to_cerber + intersect_dicts of the reference (golden: tools/make_golden_ckpt.py) -- which YOLO entry
    lands on which CerberusDet key, incl. the head copied to every task, an unknown layer and a shape mismatch."""
    import json

    from cerberusdet_amd.models import CerberusDet
    from cerberusdet_amd.utils.ckpt_utils import dict_to_cerber, intersect_dicts

    g = json.load(open(GOLDEN / "ckpt_remap.json"))[which]
    _, mmeta = load_golden("model_tiny2" if which == "tiny2" else "model_tiny3")
    m = CerberusDet(g["tasks"], g["nc"], cfg=copy.deepcopy(mmeta["cfg"]), verbose=False)  # un-split, as ModelsManager.from_ckpt sees it
    yolo = {k: torch.full(tuple(shp), float(i)) for i, (k, shp) in enumerate(zip(g["yolo_keys"], g["yolo_shapes"]))}
    mapped = dict_to_cerber(yolo, m)
    assert {k: int(v.flatten()[0]) for k, v in mapped.items()} == g["mapped"]
    final = intersect_dicts(mapped, m.state_dict(), exclude=["anchor"])
    assert {k: int(v.flatten()[0]) for k, v in final.items()} == g["final"]
    missing, unexpected = m.load_state_dict(final, strict=False)
    assert not unexpected


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Argument validation happens before any HIP call: error code < 0 and a message from cdet_last_error() (same wording as the
    reference's assertions where it has them, utils/general.py:404-405)."""
    import ctypes as C

    from cerberusdet_amd import _lib as L

    lib = L.load()
    d = L.NmsDesc()
    d.N, d.nc, d.A, d.dtype, d.conf_thres, d.iou_thres, d.max_det, d.max_nms, d.max_cand = 1, 4, 64, L.F32, 0.25, 2.0, 300, 30000, 64
    one = C.c_void_p(16)  # never dereferenced: validation fails first
    rc = lib.cdet_nms_batched(C.byref(d), one, one, one, one, None)
    assert rc < 0 and "Invalid IoU" in lib.cdet_last_error().decode()
    d.iou_thres, d.conf_thres = 0.45, -0.5
    rc = lib.cdet_nms_batched(C.byref(d), one, one, one, one, None)
    assert rc < 0 and "Invalid Confidence threshold" in lib.cdet_last_error().decode()
    cd = L.ConvDesc()
    rc = lib.cdet_conv2d(C.byref(cd), None, None, None, None, None, None, None, None)
    assert rc < 0 and "null pointer" in lib.cdet_last_error().decode()
    cd.N, cd.Hs, cd.Ws, cd.Cs, cd.Hd, cd.Wd, cd.Cd, cd.kh, cd.kw, cd.stride, cd.pad = 1, 8, 8, 12, 8, 8, 16, 3, 3, 1, 1
    cd.dtype, cd.out_dtype, cd.src_ld, cd.dst_ld = L.BF16, L.BF16, 12, 16
    rc = lib.cdet_conv2d(C.byref(cd), one, one, None, None, None, one, None, None)
    assert rc < 0 and "multiples of 8" in lib.cdet_last_error().decode()  # Cs = 12
    # the fp32 kernels of the full-precision path (csrc/precise.hip)
    rc = lib.cdet_split3(one, L.F32, 8, 0, 0, 0, None, None, one, one, 8, 0, 1, 4, 4, 8, None)
    assert rc < 0 and "null pointer" in lib.cdet_last_error().decode()
    rc = lib.cdet_split3(one, L.F32, 8, 0, 0, 1, None, one, one, one, 8, 0, 1, 5, 4, 8, None)
    assert rc < 0 and "even sides" in lib.cdet_last_error().decode()
    rc = lib.cdet_epilogue_f32(one, 8, 0, None, None, 7, None, 0, 0, one, None, None, None, 8, 0, 16, 8, None)
    assert rc < 0 and "activation" in lib.cdet_last_error().decode()
    rc = lib.cdet_maxpool_f32(one, 8, 0, one, None, None, None, 8, 0, 1, 4, 4, 8, 4, None)
    assert rc < 0 and "bad geometry" in lib.cdet_last_error().decode()
    rc = lib.cdet_bn_train_f32(one, 8, 0, 16, 8, one, one, 1e-3, 0.03, one, None, one, one, one, None, None, None)
    assert rc < 0 and "come together" in lib.cdet_last_error().decode()
    assert lib.cdet_bn_train_f32_ws_doubles(80) == 128 * 2 * 80
    md = L.MergeDesc()
    md.N, md.T, md.max_det, md.iou_thres = 1, 9, 300, 0.8
    rc = lib.cdet_merge_tasks(C.byref(md), None, one, one, None)
    assert rc < 0 and "T <= 8" in lib.cdet_last_error().decode()


def test_tiled_kernels_refuse_buffers_their_32_bit_addressing_cannot_reach():
    """The tap-resident kernels address their sources through buffer descriptors with 32-bit byte offsets (conv_halo.hip, conv_vt.hip,
    conv_pair.hip, conv_wgrad_halo.hip). A source of 3 GiB or more must be REFUSED by the geometry check (the plan compiler then takes the
    generic kernel, which uses 64-bit addresses) -- never truncated: the descriptor's range check would zero-fill the reads silently."""
    import ctypes as C

    from cerberusdet_amd import _lib as L

    lib = L.load()
    one = C.c_void_p(16)

    def desc(N, H, W, Ci, Co, k, s, ld=None):
        d = L.ConvDesc()
        d.N, d.Hs, d.Ws, d.Cs, d.Hd, d.Wd, d.Cd = N, H, W, Ci, H // s, W // s, Co
        d.kh = d.kw = k
        d.stride, d.pad, d.mode = s, k // 2, L.CONV_FWD
        d.dtype = d.out_dtype = L.F16
        d.src_ld, d.dst_ld = ld or Ci, Co
        return d

    # 160 x 160 x 448-pitch concat buffer: 128 images = 2.9 GB (accepted), 192 images = 4.4 GB (beyond 2^32: refused)
    ok = desc(128, 160, 160, 400, 160, 1, 1, ld=448)
    big = desc(192, 160, 160, 400, 160, 1, 1, ld=448)
    assert lib.cdet_conv2d_tiled_ok(C.byref(ok)) == 1 and lib.cdet_conv2d_tiled_ok(C.byref(big)) == 0
    rc = lib.cdet_conv2d_tiled(C.byref(big), one, one, None, None, None, one, None, None)
    assert rc < 0 and "unsupported geometry" in lib.cdet_last_error().decode()
    # 5 GiB source of a 3x3 stride-1 layer, and of a stride-2 layer
    big3 = desc(1400, 160, 160, 80, 80, 3, 1)
    assert 1400 * 160 * 160 * 80 * 2 > 5 * 2 ** 30 and lib.cdet_conv2d_tiled_ok(C.byref(big3)) == 0
    s2ok, s2big = desc(128, 320, 320, 80, 160, 3, 2), desc(256, 320, 320, 80, 160, 3, 2)
    assert lib.cdet_conv2d_s2_tiled_ok(C.byref(s2ok)) == 1 and lib.cdet_conv2d_s2_tiled_ok(C.byref(s2big)) == 0
    rc = lib.cdet_conv2d_s2_tiled(C.byref(s2big), one, one, None, None, None, one, None, None)
    assert rc < 0 and "unsupported geometry" in lib.cdet_last_error().decode()
    # the tap-resident weight gradient (x and dY below 3 GiB each)
    wok, wbig = desc(32, 80, 80, 320, 320, 3, 1), desc(800, 80, 80, 320, 320, 3, 1)
    assert lib.cdet_conv2d_wgrad_ws_elems(C.byref(wok)) > 0
    assert lib.cdet_conv2d_wgrad_groupable(C.byref(wok)) == 1 and lib.cdet_conv2d_wgrad_groupable(C.byref(wbig)) == 0


def test_pending_prediction_builds_the_reference_result_dicts():
    """`PendingPrediction.result()` (host side of CerberusDetInference.predict / predict_async / predict_stream): merged rows [bs, max, 6] +
    counts -> per image the reference's list of {"box", "score", "label", "label_name", "task"} (cerberusdet_inference.py:150-186): boxes
    truncated to int like the reference's int(), the score as the fp32 value, the task found from the global label id; only the first
    `count` rows of an image are read. No GPU: the event is a stand-in."""
    from cerberusdet_amd.cerberusdet_inference import CerberusDetInference, PendingPrediction

    class _Ev:
        waited = 0

        def synchronize(self):
            self.waited += 1

        def query(self):
            return True

    names = {"voc": ["a", "b", "c"], "animals": ["x", "y"]}
    cmap, all_names = CerberusDetInference._get_categories_map(names)
    assert cmap == {"voc": {0: 0, 1: 1, 2: 2}, "animals": {0: 3, 1: 4}} and all_names == ["a", "b", "c", "x", "y"]
    task_of = ["voc"] * 3 + ["animals"] * 2
    rows = torch.zeros(2, 4, 6)
    rows[0, 0] = torch.tensor([10.0, 20.0, 110.0, 220.0, 0.75, 4.0])
    rows[0, 1] = torch.tensor([0.0, 1.0, 2.0, 3.0, 0.3333333, 1.0])
    rows[0, 2] = torch.tensor([9.0, 9.0, 9.0, 9.0, 0.9, 2.0])  # beyond the count: must not appear
    rows[1, 0] = torch.tensor([5.0, 6.0, 7.0, 8.0, 0.5, 0.0])
    ev = _Ev()
    p = PendingPrediction(rows, torch.tensor([2, 1], dtype=torch.int32), ev, all_names, task_of)
    assert p.ready()
    res = p.result()
    assert res == [[{"box": [10, 20, 110, 220], "score": 0.75, "label": 4, "label_name": "y", "task": "animals"},
                    {"box": [0, 1, 2, 3], "score": float(torch.tensor(0.3333333).item()), "label": 1, "label_name": "b", "task": "voc"}],
                   [{"box": [5, 6, 7, 8], "score": 0.5, "label": 0, "label_name": "a", "task": "voc"}]]
    assert all(isinstance(v, int) for v in res[0][0]["box"]) and isinstance(res[0][0]["label"], int) and isinstance(res[0][0]["score"], float)
    assert p.result() is res and ev.waited == 1  # built once


def test_train_entry_honours_device_and_refuses_cpu(monkeypatch):
    """reference train.py:386-388 + utils/torch_utils.py:75-100: `--device` selects the GPU ('N', 'cuda:N', 'N,M'); 'cpu' would train on the
    CPU there -- here it must raise instead of silently training on cuda:0 (BASELINE config 1's literal invocation)."""
    from cerberusdet_amd import train as T
    from cerberusdet_amd.utils import torch_utils as TU

    assert TU.parse_device("") == 0 and TU.parse_device("cuda") == 0
    assert TU.parse_device("3") == 3 and TU.parse_device("cuda:5") == 5 and TU.parse_device(" 2,3 ") == 2 and TU.parse_device(1) == 1
    with pytest.raises(RuntimeError, match="not supported"):
        TU.parse_device("cpu")
    with pytest.raises(ValueError):
        TU.parse_device("gpu0")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert TU.select_device("1") == torch.device("cuda", 1)
    with pytest.raises(RuntimeError, match="only 2 GPU"):
        TU.select_device("4")
    # the entry point itself: refused before anything touches a GPU (with and without a launcher's LOCAL_RANK)
    seen = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: seen.append(d))
    monkeypatch.setattr(T, "train", lambda hyp, opt, device: seen.append(("train", device)) or "ok")
    opt = T.parse_opt(True)
    for lr in (-1, 0):
        monkeypatch.setattr(T, "LOCAL_RANK", lr)
        opt.device = "cpu"
        with pytest.raises(RuntimeError, match="not supported"):
            T.main(opt)
    assert seen == []
    monkeypatch.setattr(T, "LOCAL_RANK", -1)
    opt.device = "1"
    assert T.main(opt) == "ok"
    assert seen == [torch.device("cuda", 1), ("train", torch.device("cuda", 1))]
