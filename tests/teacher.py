"""Teacher-forced check of a compiled TRAIN plan, launch by launch (test infrastructure).

The model-level parity tests compare whole passes of a random-weight network, which is chaotic in 16-bit storage: they cannot tell a
kernel change from a bug below O(10 %). Here every unit of the compiled launch list is checked IN ISOLATION: the engine's own input
buffers (x, gy, z, mean, invstd -- whatever the launch reads) go through an fp32 PyTorch evaluation of that single layer
(tests/torchref.py) and the result is compared with what the launch wrote (z, statistics, y, dz, gx, dW, dgamma, dbeta), so no error
is carried from one layer to the next. Tolerance: 2^-7 of the tensor's scale for 16-bit outputs (one bf16 rounding is 2^-9),
1e-3 of the scale for fp32 sums. Every variant the plan compiler picks is exercised exactly as compiled: in-place Concat slices,
overwrite vs accumulate data gradients, the grouped weight-gradient launches, residual shortcuts, upsample / copy and their adjoints.

Layer semantics follow the reference: Conv = SiLU(BN(conv)) (models/common.py:51-68), Bottleneck shortcut (107-117), C2f (174-191),
SPPF pools (230-245), Concat / Upsample (288-295), Detect's biased 1x1 projections (models/yolo.py:80-100).
"""
import torch
import torch.nn.functional as F

import torchref as R
from cerberusdet_amd import _lib as L


def _f(v):
    return v.torch().float()


def _scale(t):
    return float(t.abs().max()) + 1e-30


class Report:
    def __init__(self):
        self.rows = []   # (kind, what, block, shape, err / scale)
        self.worst = {}

    def add(self, kind, what, rec, got, want, tol, name=""):
        sc = _scale(want)
        err = float((got - want).abs().max()) / sc
        key = f"{kind}.{what}"
        if err > self.worst.get(key, (0.0, ""))[0]:
            self.worst[key] = (err, f"block {rec.get('block')} {name}")
        self.rows.append((key, rec.get("block"), name, err))
        assert err <= tol, f"{key} of block {rec.get('block')} {name}: max error {err:.3e} of the tensor scale {sc:.3e} (tolerance {tol:.3e})"

    def summary(self):
        return "; ".join(f"{k} {v[0]:.1e}" for k, v in sorted(self.worst.items()))


def _w_rounded(w, dtype):
    return w.detach().to(dtype).float()


def _stem_input(plan):
    """The stem reads the NCHW image itself (uint8 scaled by 1/255, or float), rounded to the compute dtype: its NHWC fp32 view."""
    img = plan._img
    v = img.float() * (1.0 / 255.0) if img.dtype == torch.uint8 else img.float()
    return v.to(plan.dtype).float().permute(0, 2, 3, 1).contiguous()


def _conv_name(m, x):
    return f"{x.H}x{x.W} {x.C}->{m.c2} k{m.k} s{m.s}" if x is not None else f"stem ->{m.c2}"


def check_forward(plan, rep, tol16=2.0 ** -7):
    """Every forward unit from the engine's own input buffers (call after plan.run_forward + synchronize)."""
    for rec in plan.trace:
        kind = rec["kind"]
        if kind == "conv":
            m, x, z, y, res = rec["m"], rec["x"], rec["z"], rec["y"], rec["res"]
            name = _conv_name(m, x)
            w = _w_rounded(m.conv.weight, plan.dtype)
            if x is not None and getattr(m, "_stem8", False):
                w = F.pad(w, (0, 0, 0, 0, 0, x.C - w.shape[1]))
            z_ref = R.conv_fwd(_f(x) if x is not None else _stem_input(plan), w, m.s)
            rep.add("conv", "z", rec, _f(z), z_ref, tol16, name)
            mean_ref = z_ref.mean((0, 1, 2))
            var_ref = z_ref.var((0, 1, 2), unbiased=False)
            std_ref = torch.sqrt(var_ref + m.bn.eps)
            assert float(((rec["mean"] - mean_ref).abs() / std_ref).max()) < 1e-3, f"batch mean of block {rec['block']} {name}"
            assert float((rec["invstd"] * std_ref - 1).abs().max()) < 1e-3, f"batch invstd of block {rec['block']} {name}"
            a = (_f(z) - rec["mean"]) * rec["invstd"] * m.bn.weight.detach() + m.bn.bias.detach()
            y_ref = R.silu(a)
            if res is not None:
                y_ref = y_ref + _f(res)
            rep.add("conv", "y", rec, _f(y), y_ref, tol16, name)
        elif kind == "bias":
            m, x, feat = rec["m"], rec["x"], rec["feat"]
            O = m.out_channels
            ref = R.conv_fwd(_f(x), _w_rounded(m.weight, plan.dtype), 1) + m.bias.detach().float()
            rep.add("bias", "feat", rec, feat.torch()[..., :O].float(), ref, 1e-3, f"{x.C}->{O}")
        elif kind == "up":
            src, dst = rec["src"], rec["dst"]
            ref = _f(src).repeat_interleave(2, 1).repeat_interleave(2, 2)
            assert torch.equal(_f(dst), ref), f"upsample of block {rec['block']}"
        elif kind == "copy":
            assert torch.equal(_f(rec["dst"]), _f(rec["src"])), f"concat copy of block {rec['block']}"
        elif kind == "pool":
            buf, c = rec["buf"], rec["c"]
            t = buf.torch()
            cur = t[..., :c].float().permute(0, 3, 1, 2)
            for i in range(1, 4):
                cur = F.max_pool2d(cur, 5, 1, 2)
                assert torch.equal(t[..., i * c:(i + 1) * c].float(), cur.permute(0, 2, 3, 1)), f"SPPF pool {i}"


def _snap(rec, plan):
    s = {}
    kind = rec["kind"]
    if kind == "conv":
        m = rec["m"]
        if rec["gx"] is not None and rec["gx_acc"]:
            s["gx"] = rec["gx"].torch().clone()
        s["gw"], s["gb"] = m.bn.weight.grad.clone(), m.bn.bias.grad.clone()
        if rec.get("also") is not None:
            s["also"] = rec["also"].torch().clone()
        if not rec["grouped"]:
            s["cw"] = m.conv.weight.grad.clone()
    elif kind == "bias":
        m = rec["m"]
        if rec["gx_acc"]:
            s["gx"] = rec["gx"].torch().clone()
        s["cw"], s["cb"] = m.weight.grad.clone(), m.bias.grad.clone()
    elif kind in ("up", "copy"):
        if rec["acc"]:
            s["gs"] = rec["gs"].torch().clone()
    return s


def _dz_view(rec):
    z = rec["z"]
    dz = rec["dz"]
    return dz.reshape(-1)[:z.M * z.C].view(z.N, z.H, z.W, z.C)


def _check_bwd(rec, s, plan, rep, tol16, stash):
    kind = rec["kind"]
    if kind == "conv":
        m, x, z = rec["m"], rec["x"], rec["z"]
        name = _conv_name(m, x)
        bn = m.bn
        g, be = bn.weight.detach(), bn.bias.detach()
        gy, zf = _f(rec["gy"]), _f(z)
        xh = (zf - rec["mean"]) * rec["invstd"]
        dU = gy * R.dsilu(g * xh + be)
        db_ref = dU.sum((0, 1, 2))
        dg_ref = (dU * xh).sum((0, 1, 2))
        M = z.M
        dz_ref = g * rec["invstd"] * (dU - db_ref / M - xh * (dg_ref / M))
        dz_e = _dz_view(rec).float()
        rep.add("conv", "dz", rec, dz_e, dz_ref, tol16, name)
        rep.add("conv", "dgamma", rec, bn.weight.grad - s["gw"], dg_ref, 2e-3, name)
        rep.add("conv", "dbeta", rec, bn.bias.grad - s["gb"], db_ref, 2e-3, name)
        if rec.get("also") is not None:  # Bottleneck shortcut: the same pass adds grad(y) into the other addend's gradient slice
            rep.add("conv", "shortcut", rec, _f(rec["also"]), s["also"].float() + gy, tol16, name)
        w = _w_rounded(m.conv.weight, plan.dtype)
        if rec["gx"] is not None:
            ref = R.conv_dgrad(dz_e, w, m.s, x.H, x.W)
            if rec["gx_acc"]:
                ref = ref + s["gx"].float()
            rep.add("conv", "gx+" if rec["gx_acc"] else "gx", rec, _f(rec["gx"]), ref, tol16, name)
        dw_ref = R.conv_wgrad(_f(x) if x is not None else _stem_input(plan), dz_e, m.k, m.s)[:, :m.conv.weight.shape[1]]
        if rec["grouped"]:
            stash.append((rec, dw_ref, name))
        else:
            rep.add("conv", "dw", rec, m.conv.weight.grad - s["cw"], dw_ref, 1e-3, name)
    elif kind == "bias":
        m, x, dfeat = rec["m"], rec["x"], rec["dfeat"]
        O = m.out_channels
        d = dfeat.torch()[..., :O].float()
        w = _w_rounded(m.weight, plan.dtype)
        ref = R.conv_dgrad(d, w, 1, x.H, x.W)
        if rec["gx_acc"]:
            ref = ref + s["gx"].float()
        rep.add("bias", "gx+" if rec["gx_acc"] else "gx", rec, _f(rec["gx"]), ref, tol16, f"{x.C}->{O}")
        rep.add("bias", "dw", rec, m.weight.grad - s["cw"], R.conv_wgrad(_f(x), d, 1, 1), 1e-3, f"{x.C}->{O}")
        rep.add("bias", "db", rec, m.bias.grad - s["cb"], d.sum((0, 1, 2)), 1e-3, f"{x.C}->{O}")
    elif kind == "up":
        gd = _f(rec["gd"])
        N, H2, W2, Cn = gd.shape
        ref = gd.view(N, H2 // 2, 2, W2 // 2, 2, Cn).sum((2, 4))
        if rec["acc"]:
            ref = ref + s["gs"].float()
        rep.add("up", "gs", rec, _f(rec["gs"]), ref, tol16)
    elif kind == "copy":
        ref = _f(rec["gd"])
        if rec["acc"]:
            ref = ref + s["gs"].float()
        rep.add("copy", "gs", rec, _f(rec["gs"]), ref, tol16)


def check_backward(plan, rep, tol16=2.0 ** -7):
    """Replays the plan's backward launch list call by call (the head-map gradients must already sit in plan.dfeats) and checks every
    unit's writes against the fp32 evaluation of that unit from the buffers it read."""
    plan.attach_grads()
    recs = [r for r in plan.trace if "bwd_lo" in r]
    starts = {(r["bwd_group"], r["bwd_lo"]): r for r in recs}
    ends = {}
    for r in recs:
        ends.setdefault((r["bwd_group"], r["bwd_hi"]), []).append(r)
    assert len(starts) == len(recs)
    st = torch.cuda.current_stream().cuda_stream
    n_checked = 0
    for gi, (idx, calls) in enumerate(plan.bwd_groups):
        grouped = [r for r in recs if r["kind"] == "conv" and r["bwd_group"] == gi and r["grouped"]]
        cw0 = {id(r): r["m"].conv.weight.grad.clone() for r in grouped}
        stash = []
        snaps = {}
        for ci, (fn, args) in enumerate(calls):
            r = starts.get((gi, ci))
            if r is not None:
                snaps[id(r)] = _snap(r, plan)
            rc = fn(*args, st)
            if rc:
                L.check(rc, getattr(fn, "__name__", "call"))
            for r in ends.get((gi, ci + 1), []):
                _check_bwd(r, snaps.pop(id(r)), plan, rep, tol16, stash)
                n_checked += 1
        assert not snaps, "a unit's launches ran past the end of its block's group"
        assert len(stash) == len(grouped)
        for r, dw_ref, name in stash:  # the block's grouped weight-gradient launches ran at the end of the group
            rep.add("conv", "dw(grouped)", r, r["m"].conv.weight.grad - cw0[id(r)], dw_ref, 1e-3, name)
    assert n_checked == len(recs)
    return n_checked


# ---------------------------------------------------------------------------------------------------------------------------------------------
# EVAL plans (round 5): the launch list the north-star forward and CerberusDetInference run -- folded-BatchNorm epilogues, virtual Concat /
# Upsample sources, the fused first two backbone rows, fp32 projections, the one-launch SPPF pool chain, decode.
# ---------------------------------------------------------------------------------------------------------------------------------------------
def _cat_input(x):
    """fp32 NHWC tensor of a unit's input: a View, or the parts of a virtual Concat (an upsampled part is read through (y/2, x/2),
    reference nn.Upsample(nearest, 2) + Concat, models/common.py:288-295)."""
    if hasattr(x, "parts"):
        ts = []
        for v, up in x.parts:
            t = _f(v)
            ts.append(t.repeat_interleave(2, 1).repeat_interleave(2, 2) if up else t)
        return torch.cat(ts, 3)
    return _f(x)


def _folded(m):
    """(scale | None, bias) of the eval-form epilogue, as the engine packed them (engine._pack_conv: BatchNorm folded, reference
    utils/torch_utils.py:191-217), recomputed here from the module's own state."""
    if m.fused:
        return None, m.conv.bias.detach().float()
    bn = m.bn
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
    return scale, (bn.bias - bn.running_mean * scale).detach().float()


def _eval_act(z, m):
    scale, bias = _folded(m)
    return R.silu(z * scale + bias if scale is not None else z + bias)


def decode_ref(feats, nc, strides):
    """Eval branch of Detect.forward incl. DFL (reference models/yolo.py:57-59, 93-100; utils/tal.py:181-205) on the padded NHWC fp32 head maps
    [N, h, w, 64 + pad8(nc)] -> y [N, 4 + nc, A] fp32."""
    boxes, clss, pts, sts = [], [], [], []
    for f, s in zip(feats, strides):
        N, h, w, _ = f.shape
        boxes.append(f[..., :64].reshape(N, h * w, 4, 16))
        clss.append(f[..., 64:64 + nc].reshape(N, h * w, nc))
        yy, xx = torch.meshgrid(torch.arange(h, device=f.device, dtype=torch.float32) + 0.5,
                                torch.arange(w, device=f.device, dtype=torch.float32) + 0.5, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        sts.append(torch.full((h * w, 1), float(s), device=f.device))
    box, cls, pt, st = torch.cat(boxes, 1), torch.cat(clss, 1), torch.cat(pts), torch.cat(sts)
    dist = (box.softmax(3) * torch.arange(16, device=box.device, dtype=torch.float32)).sum(3)  # [N, A, 4] ltrb
    x1y1, x2y2 = pt - dist[..., :2], pt + dist[..., 2:]
    dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 2) * st
    return torch.cat((dbox, cls.sigmoid()), 2).permute(0, 2, 1)


def check_eval_forward(plan, rep, tol16=2.0 ** -7):
    """Every unit of a compiled EVAL plan from the engine's own input buffers (call after plan.run_forward + synchronize; the plan must write its
    home outputs: plan.fresh_outputs(False)). Returns the number of units checked per kind."""
    assert not plan.training
    n = {}
    for rec in plan.trace:
        kind = rec["kind"]
        n[kind] = n.get(kind, 0) + 1
        if kind == "econv":
            m, x, y, res = rec["m"], rec["x"], rec["y"], rec["res"]
            w = _w_rounded(m.conv.weight, plan.dtype)
            if x is None:
                xin, name = _stem_input(plan), f"stem ->{m.c2}"
            else:
                xin = _cat_input(x)
                name = f"{y.H * m.s}x{y.W * m.s} {xin.shape[3]}->{m.c2} k{m.k} s{m.s}" + (" [virtual cat]" if hasattr(x, "parts") else "")
            ref = _eval_act(R.conv_fwd(xin, w, m.s), m)
            if res is not None:
                ref = ref + _f(res)
            rep.add("econv", "y", rec, _f(y), ref, tol16, name)
        elif kind == "stemc1":
            m0, m1, y = rec["m0"], rec["m1"], rec["y"]
            s0 = _eval_act(R.conv_fwd(_stem_input(plan), _w_rounded(m0.conv.weight, plan.dtype), m0.s), m0)
            s0 = s0.to(plan.dtype).float()  # the stem's map lives in LDS in the compute dtype (csrc/stem_conv1.hip), as the two-kernel path stores it
            ref = _eval_act(R.conv_fwd(s0, _w_rounded(m1.conv.weight, plan.dtype), m1.s), m1)
            rep.add("stemc1", "y", rec, _f(y), ref, tol16, f"3->{m0.c2}->{m1.c2}")
        elif kind == "ebias":
            m, x, feat = rec["m"], rec["x"], rec["feat"]
            O = m.out_channels
            ref = R.conv_fwd(_f(x), _w_rounded(m.weight, plan.dtype), 1) + m.bias.detach().float()
            rep.add("ebias", "feat", rec, feat.torch()[..., :O].float(), ref, 1e-3, f"{x.H}x{x.W} {x.C}->{O}")
        elif kind == "epool":
            buf, c = rec["buf"], rec["c"]
            t = buf.torch()
            cur = t[..., :c].float().permute(0, 3, 1, 2)
            for i in range(1, 4):
                cur = F.max_pool2d(cur, 5, 1, 2)
                assert torch.equal(t[..., i * c:(i + 1) * c].float(), cur.permute(0, 2, 3, 1)), f"SPPF pool {i}"
        elif kind == "decode":
            head, feats, y = rec["head"], rec["feats"], rec["y"]
            assert all(f.data_ptr() == g.data_ptr() for f, g in zip(feats, plan.feats[rec["task"]])), "check the plan's home outputs (fresh_outputs(False))"
            ref = decode_ref(feats, head.nc, [float(s) for s in head.stride])
            got = y.float()
            tol = 1e-3 if y.dtype == torch.float32 else 2.0 ** -9  # (a 16-bit `y` rounds once: fp16 2^-11, bf16 2^-8 of the value)
            # boxes against the image scale, class probabilities absolute
            rep.add("decode", "box", rec, got[:, :4], ref[:, :4], tol if y.dtype != torch.bfloat16 else 2.0 ** -7, rec["task"])
            assert float((got[:, 4:] - ref[:, 4:]).abs().max()) <= (1e-3 if y.dtype != torch.bfloat16 else 2.0 ** -8), f"class probabilities of {rec['task']}"
    return n
