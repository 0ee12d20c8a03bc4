"""fp32-accurate execution of a CerberusDet eval forward THROUGH THE HIP CONVOLUTION KERNELS (test infrastructure).

The product path stores activations and GEMM operands in bf16, which is a statistical match to the fp32 reference. To check the
kernels and the weight / BatchNorm handling at the reference's own precision, every convolution here is evaluated as three bf16
MFMA convolutions accumulated in fp32 (operand splitting: x = x_hi + x_lo, w = w_hi + w_lo with 8-bit mantissas each;
x_hi*w_hi + x_hi*w_lo + x_lo*w_hi leaves a relative error of about 2^-16), followed by the folded BatchNorm + SiLU in fp32.
Plumbing (concat, upsample, max-pool) uses torch ops -- they are exact. The Detect decode runs on the product kernel.
"""
import torch
import torch.nn.functional as F

from cerberusdet_amd import _lib as L
from cerberusdet_amd import ops
from cerberusdet_amd.models.common import C2f, Concat, Conv, SPPF, Upsample


def _split(t32):
    hi = t32.to(torch.bfloat16)
    lo = (t32 - hi.float()).to(torch.bfloat16)
    return hi, lo


def conv3(x, w, k, s):
    """x [N,C,H,W] fp32 (cuda), w [O,I,k,k] fp32 -> raw conv [N,O,Ho,Wo] fp32 via 3 bf16 MFMA convs with fp32 accumulation."""
    N, Ci, H, W = x.shape
    O = w.shape[0]
    Cp, Op = (Ci + 7) // 8 * 8, (O + 7) // 8 * 8
    xn = torch.zeros((N, H, W, Cp), dtype=torch.float32, device=x.device)
    xn[..., :Ci] = x.permute(0, 2, 3, 1)
    wp = torch.zeros((Op, Cp, k, k), dtype=torch.float32, device=x.device)
    wp[:O, :Ci] = w
    xh, xl = _split(xn)
    wh, wl = _split(wp)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    y = ops.new_act(N, Ho, Wo, Op, torch.float32)
    ph, pl = ops.pack_weight(wh.float(), torch.bfloat16), ops.pack_weight(wl.float(), torch.bfloat16)
    ops.conv2d(ops.View(xh.contiguous()), ph, y, k, s)
    ops.conv2d(ops.View(xh.contiguous()), pl, y, k, s, accumulate=True)
    ops.conv2d(ops.View(xl.contiguous()), ph, y, k, s, accumulate=True)
    return y.buf[..., :O].permute(0, 3, 1, 2).contiguous()


def conv_unit(m: Conv, x):
    if getattr(m, "fused", False):
        return F.silu(conv3(x, m.conv.weight.detach().float(), m.k, m.s) + m.conv.bias.detach().float().view(1, -1, 1, 1))
    bn = m.bn
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
    bias = (bn.bias - bn.running_mean * scale).detach().float()
    return F.silu(conv3(x, m.conv.weight.detach().float(), m.k, m.s) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))


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


@torch.no_grad()
def eval_forward(model, img, tasks=None):
    """-> {task: (y [N,4+nc,A] fp32, [3 maps [N,64+nc,h,w]])} like model.eval()(img)."""
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
                    parts.append(conv3(t, br[2].weight.detach().float(), 1, 1) + br[2].bias.detach().float().view(1, -1, 1, 1))
                mp = torch.cat(parts, 1)
                maps.append(mp)
                fb = torch.zeros((mp.shape[0], mp.shape[2], mp.shape[3], 64 + ncp), dtype=torch.float32, device=mp.device)
                fb[..., :64 + blk.nc] = mp.permute(0, 2, 3, 1)
                feats.append(fb)
            y = ops.detect_decode(feats, blk.nc, [float(s) for s in blk.stride])
            res[task] = (y, maps)
        else:
            outs[idx] = layer(blk, xs)
    return res
