"""fp32-accurate execution of a CerberusDet eval forward THROUGH THE HIP CONVOLUTION KERNELS (test infrastructure).

The product path stores activations and GEMM operands in bf16, which is a statistical match to the fp32 reference. To check the
kernels and the weight / BatchNorm handling at the reference's own precision, every convolution here is evaluated as six bf16
MFMA convolutions (round 4: on the tap-resident kernels that carry the product -- conv_halo.hip / conv_vt.hip with their fp32
accumulate epilogue -- wherever they take the geometry) accumulated in fp32 (operand splitting: x = x_hi + x_mid + x_lo, likewise w, 8 mantissa bits per term; the six
term pairs above 2^-24 are kept), followed by the folded (or batch-statistics) BatchNorm + SiLU in fp32.
Plumbing (concat, upsample, max-pool) uses torch ops -- they are exact. The Detect decode runs on the product kernel.
"""
import torch
import torch.nn.functional as F

from cerberusdet_amd import _lib as L
from cerberusdet_amd import ops
from cerberusdet_amd.models.common import C2f, Concat, Conv, SPPF, Upsample


def _split(t32):
    """fp32 -> three bf16 terms (8 + 8 + 8 mantissa bits): t = hi + mid + lo up to 2^-24."""
    hi = t32.to(torch.bfloat16)
    r = t32 - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi.contiguous(), mid.contiguous(), lo.contiguous()


# operand-term pairs whose product is above 2^-24 relative: (0,0) (0,1) (1,0) (1,1) (0,2) (2,0)
_PAIRS = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]

# which entry point carried each convolution / data gradient (round 4: the tap-resident kernels write and accumulate fp32, so the
# accuracy chain runs through the kernels that carry the product -- cdet_conv2d_tiled / cdet_conv2d_s2_tiled[_dgrad]; the generic
# cdet_conv2d only takes the geometries those refuse)
CALLS = {"tiled": 0, "s2_tiled": 0, "generic": 0, "tiled_dgrad": 0, "s2_tiled_dgrad": 0, "generic_dgrad": 0}
FORCE_GENERIC = False


def conv3(x, w, k, s):
    """x [N,C,H,W] fp32 (cuda), w [O,I,k,k] fp32 -> raw conv [N,O,Ho,Wo] fp32 via 3 bf16 MFMA convs with fp32 accumulation."""
    N, Ci, H, W = x.shape
    O = w.shape[0]
    Cp, Op = (Ci + 7) // 8 * 8, (O + 7) // 8 * 8
    xn = torch.zeros((N, H, W, Cp), dtype=torch.float32, device=x.device)
    xn[..., :Ci] = x.permute(0, 2, 3, 1)
    wp = torch.zeros((Op, Cp, k, k), dtype=torch.float32, device=x.device)
    wp[:O, :Ci] = w
    xs = _split(xn)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    y = ops.new_act(N, Ho, Wo, Op, torch.float32)
    xv = ops.View(xs[0])
    path = "generic"
    if not FORCE_GENERIC:
        if s == 1 and ops.conv2d_tiled_ok(xv, y, k, 1, accumulate=True):
            path = "tiled"
        elif s == 2 and k == 3 and ops.conv2d_s2_tiled_ok(xv, y):
            path = "s2_tiled"
    CALLS[path] += 1
    if path == "generic":
        wpk = [ops.pack_weight(t.float(), torch.bfloat16) for t in _split(wp)]
    else:
        wpk = [ops.pack_weight_tiled(t.float(), torch.bfloat16, fwd=True, dgrad=False)[0] for t in _split(wp)]
    for n_, (i, j) in enumerate(_PAIRS):
        if path == "tiled":
            ops.conv2d_tiled(ops.View(xs[i]), wpk[j], y, k, accumulate=n_ > 0)
        elif path == "s2_tiled":
            ops.conv2d_s2_tiled(ops.View(xs[i]), wpk[j], y, accumulate=n_ > 0)
        else:
            ops.conv2d(ops.View(xs[i]), wpk[j], y, k, s, accumulate=n_ > 0)
    return y.buf[..., :O].permute(0, 3, 1, 2).contiguous()


def dgrad3(dy, w, k, s, x_shape):
    """dX of conv3: three bf16 data-gradient launches (stride 2: the parity-class kernel) accumulated in fp32."""
    N, Ci, H, W = x_shape
    O = w.shape[0]
    Cp, Op = (Ci + 7) // 8 * 8, (O + 7) // 8 * 8
    dn = torch.zeros((N, dy.shape[2], dy.shape[3], Op), dtype=torch.float32, device=dy.device)
    dn[..., :O] = dy.permute(0, 2, 3, 1)
    wp = torch.zeros((O, Cp, k, k), dtype=torch.float32, device=dy.device)
    wp[:, :Ci] = w
    ds = _split(dn)
    dx = ops.new_act(N, H, W, Cp, torch.float32)
    dv = ops.View(ds[0])
    path = "generic_dgrad"
    if not FORCE_GENERIC:
        if s == 1 and ops.conv2d_tiled_ok(dv, dx, k, 1, accumulate=True):
            path = "tiled_dgrad"  # the stride-1 data gradient = forward convolution of dY with the DGRAD operand
        elif s == 2 and k == 3 and ops.conv2d_s2_tiled_ok(dv, dx, L.CONV_DGRAD):
            path = "s2_tiled_dgrad"
    CALLS[path] += 1
    if path == "generic_dgrad":
        wtk = [ops.pack_weight(t.float(), torch.bfloat16, transpose=True, o_pad=Op) for t in _split(wp)]
    else:
        wpp = torch.zeros((Op, Cp, k, k), dtype=torch.float32, device=dy.device)
        wpp[:O] = wp
        wtk = [ops.pack_weight_tiled(t.float(), torch.bfloat16, fwd=False, dgrad=True)[1] for t in _split(wpp)]
    for n_, (i, j) in enumerate(_PAIRS):
        if path == "tiled_dgrad":
            ops.conv2d_tiled(ops.View(ds[i]), wtk[j], dx, k, accumulate=n_ > 0)
        elif path == "s2_tiled_dgrad":
            ops.conv2d_s2_tiled_dgrad(ops.View(ds[i]), wtk[j], dx, accumulate=n_ > 0)
        else:
            ops.conv2d(ops.View(ds[i]), wtk[j], dx, k, s, mode=L.CONV_DGRAD, accumulate=n_ > 0)
    return dx.buf[..., :Ci].permute(0, 3, 1, 2).contiguous()


def wgrad3(x, dy, k, s, w_shape):
    """dW of conv3: three bf16 weight-gradient launches accumulated in fp32."""
    O, Ci = w_shape[0], w_shape[1]
    N, _, H, W = x.shape
    Cp, Op = (Ci + 7) // 8 * 8, (O + 7) // 8 * 8
    xn = torch.zeros((N, H, W, Cp), dtype=torch.float32, device=x.device)
    xn[..., :Ci] = x.permute(0, 2, 3, 1)
    dn = torch.zeros((N, dy.shape[2], dy.shape[3], Op), dtype=torch.float32, device=x.device)
    dn[..., :O] = dy.permute(0, 2, 3, 1)
    xs, ds = _split(xn), _split(dn)
    dw = torch.zeros((Op, Cp, k, k), dtype=torch.float32, device=x.device)
    for n_, (i, j) in enumerate(_PAIRS):
        ops.conv2d_wgrad(ops.View(xs[i]), ops.View(ds[j]), dw, k, s, accumulate=n_ > 0)
    return dw[:O, :Ci].contiguous()


class Conv3Fn(torch.autograd.Function):
    """conv3 with its gradients on the HIP data-gradient / weight-gradient kernels (everything around it is torch autograd in fp32)."""

    @staticmethod
    def forward(ctx, x, w, k, s):
        ctx.save_for_backward(x, w)
        ctx.ks = (k, s)
        return conv3(x, w, k, s)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, s = ctx.ks
        dy = dy.contiguous()
        dx = dgrad3(dy, w, k, s, x.shape) if ctx.needs_input_grad[0] else None
        dw = wgrad3(x, dy, k, s, w.shape) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


_PARAMS = None  # {id(model parameter): fp32 leaf used by the autograd run}; None -> detached parameters (eval forward)


def _p(t):
    return t.detach().float() if _PARAMS is None else _PARAMS[id(t)]


def _conv(x, w, k, s):
    return conv3(x, w, k, s) if _PARAMS is None else Conv3Fn.apply(x, w, k, s)


def conv_unit(m: Conv, x, training=False):
    if getattr(m, "fused", False):
        return F.silu(_conv(x, _p(m.conv.weight), m.k, m.s) + _p(m.conv.bias).view(1, -1, 1, 1))
    bn = m.bn
    z = _conv(x, _p(m.conv.weight), m.k, m.s)
    if _TRAIN:
        return F.silu(F.batch_norm(z, None, None, _p(bn.weight), _p(bn.bias), True, 0.0, bn.eps))
    scale = _p(bn.weight) / torch.sqrt(bn.running_var.float() + bn.eps)
    bias = _p(bn.bias) - bn.running_mean.float() * scale
    return F.silu(z * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))


_TRAIN = False


def layer(m, xs):
    if isinstance(m, Conv):
        return conv_unit(m, xs[0])
    if isinstance(m, C2f):
        y = list(conv_unit(m.cv1, xs[0]).chunk(2, 1))
        for b in m.m:
            t = conv_unit(b.cv2, conv_unit(b.cv1, y[-1]))
            y.append(y[-1] + t if b.add else t)
        return conv_unit(m.cv2, torch.cat(y, 1))
    if isinstance(m, SPPF):
        x = conv_unit(m.cv1, xs[0])
        y1 = F.max_pool2d(x, m.k, 1, m.k // 2)
        y2 = F.max_pool2d(y1, m.k, 1, m.k // 2)
        return conv_unit(m.cv2, torch.cat((x, y1, y2, F.max_pool2d(y2, m.k, 1, m.k // 2)), 1))
    if isinstance(m, Upsample):
        return F.interpolate(xs[0], scale_factor=2, mode="nearest")
    if isinstance(m, Concat):
        return torch.cat(xs, 1)
    raise NotImplementedError(type(m))


def train_forward(model, img, task, img_grad=False):
    """Train-mode forward of one task (BatchNorm from batch statistics) as a torch autograd graph whose convolutions run on the HIP
    kernels. Returns (maps, params) with params = {state-dict key: fp32 leaf}; call backward on the maps, read params[k].grad."""
    global _PARAMS, _TRAIN
    named = dict(model.named_parameters())
    leaves = {k: p.detach().clone().float().requires_grad_(True) for k, p in named.items()}
    _PARAMS, _TRAIN = {id(named[k]): leaves[k] for k in named}, True
    try:
        with torch.enable_grad():
            if img_grad:
                img = img.detach().float().requires_grad_(True)
                leaves["__img__"] = img
            res = _forward(model, img, [task], decode=False)
    finally:
        _PARAMS, _TRAIN = None, False
    return res[task][1], leaves


@torch.no_grad()
def eval_forward(model, img, tasks=None):
    """-> {task: (y [N,4+nc,A] fp32, [3 maps [N,64+nc,h,w]])} like model.eval()(img)."""
    return _forward(model, img, tasks, decode=True)


def _forward(model, img, tasks, decode):
    tasks = list(model.heads) if tasks is None else tasks
    x = img.float() / 255.0 if img.dtype == torch.uint8 else img.float()
    order, _ = model.execution_plan(tasks)
    outs, res = {}, {}
    for idx in order:
        blk = model.blocks[idx]
        if idx == 0:
            ys, cur = [], x
            for li, lay in enumerate(blk.model):
                f = lay.f
                xin = [cur] if (li == 0 or f == -1) else ([ys[f]] if isinstance(f, int) else [cur if j == -1 else ys[j] for j in f])
                cur = layer(lay, xin)
                ys.append(cur)
            outs[0] = ys
            continue
        xs = [outs[0][j] if kind == "bb" else outs[j] for kind, j in model._inputs[idx]]
        if idx in model.heads.values():
            task = model.controllers[idx].task_id
            maps, feats = [], []
            ncp = (blk.nc + 7) // 8 * 8
            for lvl, xl in enumerate(xs):
                parts = []
                for br in (blk.cv2[lvl], blk.cv3[lvl]):
                    t = conv_unit(br[1], conv_unit(br[0], xl))
                    parts.append(_conv(t, _p(br[2].weight), 1, 1) + _p(br[2].bias).view(1, -1, 1, 1))
                mp = torch.cat(parts, 1)
                maps.append(mp)
                if decode:
                    fb = torch.zeros((mp.shape[0], mp.shape[2], mp.shape[3], 64 + ncp), dtype=torch.float32, device=mp.device)
                    fb[..., :64 + blk.nc] = mp.permute(0, 2, 3, 1)
                    feats.append(fb)
            y = ops.detect_decode(feats, blk.nc, [float(s) for s in blk.stride]) if decode else None
            res[task] = (y, maps)
        else:
            outs[idx] = layer(blk, xs)
    return res
