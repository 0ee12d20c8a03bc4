"""Thin host wrappers: torch tensors own the memory, libcerberus_hip.so does the work.

NHWC activations are described by `View` = (buffer [N,H,W,LD], channel offset, channel count) so that producers
can write straight into channel slices of concat buffers (models/common.py:191,245,295 never materialise a cat).
Every wrapper launches asynchronously on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

_DT = {torch.bfloat16: L.BF16, torch.float16: L.F16, torch.float32: L.F32, torch.uint8: L.U8}


def dt(t) -> int:
    return _DT[t if isinstance(t, torch.dtype) else t.dtype]


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> Optional[int]:
    if t is None:
        return None
    if isinstance(t, View):
        return t.buf.data_ptr()
    return t.data_ptr()


class View:
    """Channel slice [coff, coff+C) of a contiguous NHWC buffer."""

    __slots__ = ("buf", "coff", "C")

    def __init__(self, buf: torch.Tensor, coff: int = 0, C: Optional[int] = None):
        assert buf.dim() == 4 and buf.is_contiguous()
        self.buf, self.coff = buf, coff
        self.C = buf.shape[3] - coff if C is None else C

    N = property(lambda s: s.buf.shape[0])
    H = property(lambda s: s.buf.shape[1])
    W = property(lambda s: s.buf.shape[2])
    ld = property(lambda s: s.buf.shape[3])
    M = property(lambda s: s.buf.shape[0] * s.buf.shape[1] * s.buf.shape[2])
    dtype = property(lambda s: s.buf.dtype)

    def slice(self, c0, c):
        return View(self.buf, self.coff + c0, c)

    def torch(self):
        return self.buf[..., self.coff:self.coff + self.C]

    def nchw(self):
        return self.torch().permute(0, 3, 1, 2)


def new_act(N, H, W, C, dtype, device="cuda", zero=False):
    f = torch.zeros if zero else torch.empty
    return View(f((N, H, W, C), dtype=dtype, device=device))


def from_nchw(x: torch.Tensor, dtype=None) -> View:
    x = x.permute(0, 2, 3, 1).contiguous()
    return View(x.to(dtype) if dtype is not None else x)


# ---------------------------------------------------------------------------------------------------------------
def conv_desc(src: View, dst: View, k, s, mode=L.CONV_FWD, act=L.ACT_NONE, res: Optional[View] = None, accumulate=False,
              pad=None) -> L.ConvDesc:
    d = L.ConvDesc()
    d.N, d.Hs, d.Ws, d.Cs = src.N, src.H, src.W, src.C
    d.Hd, d.Wd, d.Cd = dst.H, dst.W, dst.C
    d.kh = d.kw = k
    d.stride, d.pad, d.mode = s, (k // 2 if pad is None else pad), mode
    d.dtype, d.out_dtype, d.act = dt(src.dtype), dt(dst.dtype), act
    d.src_ld, d.src_coff, d.dst_ld, d.dst_coff = src.ld, src.coff, dst.ld, dst.coff
    d.res_ld, d.res_coff = (res.ld, res.coff) if res is not None else (0, 0)
    d.accumulate = 1 if accumulate else 0
    return d


def pack_weight(w_oihw: torch.Tensor, dtype, transpose=False, row_scale=None, o_pad=None) -> torch.Tensor:
    lib = L.load()
    O, I, kh, kw = w_oihw.shape
    o_pad = O if o_pad is None else o_pad
    w32 = w_oihw.detach().float().contiguous()
    n = lib.cdet_packed_weight_elems(o_pad, I, kh, kw, int(transpose))
    out = torch.empty(n, dtype=dtype, device=w_oihw.device)
    L.check(lib.cdet_pack_weight(ptr(w32), ptr(out), O, o_pad, I, kh, kw, int(transpose), ptr(row_scale), dt(dtype), stream()),
            "cdet_pack_weight")
    return out


def conv2d(src: View, w_packed, dst: View, k, s, scale=None, bias=None, act=L.ACT_NONE, res: Optional[View] = None,
           stats=None, mode=L.CONV_FWD, accumulate=False, desc=None):
    lib = L.load()
    d = desc or conv_desc(src, dst, k, s, mode, act, res, accumulate)
    L.check(lib.cdet_conv2d(C.byref(d), ptr(src), ptr(w_packed), ptr(scale), ptr(bias), ptr(res), ptr(dst), ptr(stats), stream()),
            "cdet_conv2d")
    return dst


def pack_weight_tiled(w_oihw: torch.Tensor, dtype, fwd=True, dgrad=False):
    """Tiled operands of the tap-resident kernel (cdet_conv2d_tiled): returns (w_fwd | None, w_dgrad | None)."""
    lib = L.load()
    O, I, kh, kw = w_oihw.shape
    w32 = w_oihw.detach().float().contiguous()
    wf = torch.zeros(lib.cdet_tiled_weight_elems(O, I, kh, kw), dtype=dtype, device=w_oihw.device) if fwd else None
    wd = torch.zeros(lib.cdet_tiled_weight_elems(I, O, kh, kw), dtype=dtype, device=w_oihw.device) if dgrad else None
    L.check(lib.cdet_pack_weight_tiled(ptr(w32), ptr(wf), ptr(wd), O, I, kh, kw, dt(dtype), stream()), "cdet_pack_weight_tiled")
    return wf, wd


def conv2d_tiled_ok(src: View, dst: View, k, s, accumulate=False) -> bool:
    d = conv_desc(src, dst, k, s, accumulate=accumulate)
    return bool(L.load().cdet_conv2d_tiled_ok(C.byref(d)))


def conv2d_tiled(src: View, w_tiled, dst: View, k, scale=None, bias=None, act=L.ACT_NONE, res: Optional[View] = None, stats=None,
                 accumulate=False):
    """dst may be fp32 (HEPI_F32 epilogue); accumulate (fp32 dst only): dst += result."""
    lib = L.load()
    d = conv_desc(src, dst, k, 1, L.CONV_FWD, act, res, accumulate)
    L.check(lib.cdet_conv2d_tiled(C.byref(d), ptr(src), ptr(w_tiled), ptr(scale), ptr(bias), ptr(res), ptr(dst), ptr(stats), stream()),
            "cdet_conv2d_tiled")
    return dst


def cat_srcs(parts):
    """parts: [(View, upsample: bool)] -> (ctypes array of cdet_cat_src, total channels)."""
    arr = (L.CatSrc * len(parts))()
    for i, (v, up) in enumerate(parts):
        arr[i].x, arr[i].ld, arr[i].coff, arr[i].C, arr[i].upsample = v.buf.data_ptr(), v.ld, v.coff, v.C, int(bool(up))
    return arr, sum(v.C for v, _ in parts)


def cat_desc(parts, dst: View, act=L.ACT_NONE, res: Optional[View] = None) -> L.ConvDesc:
    """Descriptor of a 1x1 convolution over the virtual Concat of `parts` (an upsampled part is half the destination's size)."""
    d = L.ConvDesc()
    d.N, d.Hs, d.Ws, d.Cs = dst.N, dst.H, dst.W, sum(v.C for v, _ in parts)
    d.Hd, d.Wd, d.Cd = dst.H, dst.W, dst.C
    d.kh = d.kw = 1
    d.stride, d.pad, d.mode = 1, 0, L.CONV_FWD
    d.dtype, d.out_dtype, d.act = dt(parts[0][0].dtype), dt(dst.dtype), act
    d.src_ld, d.src_coff, d.dst_ld, d.dst_coff = d.Cs, 0, dst.ld, dst.coff
    d.res_ld, d.res_coff = (res.ld, res.coff) if res is not None else (0, 0)
    return d


def conv2d_tiled_cat(parts, w_tiled, dst: View, scale=None, bias=None, act=L.ACT_NONE, res: Optional[View] = None):
    """1x1 convolution over a virtual Concat (+ nearest 2x Upsample) of up to three NHWC views (csrc/conv_halo.hip / conv_pair.hip, CAT forms)."""
    lib = L.load()
    arr, _ = cat_srcs(parts)
    d = cat_desc(parts, dst, act, res)
    L.check(lib.cdet_conv2d_tiled_cat(C.byref(d), arr, len(parts), ptr(w_tiled), ptr(scale), ptr(bias), ptr(res), ptr(dst), stream()), "cdet_conv2d_tiled_cat")
    return dst


def conv2d_s2_tiled_ok(src: View, dst: View, mode=L.CONV_FWD) -> bool:
    d = conv_desc(src, dst, 3, 2, mode)
    return bool(L.load().cdet_conv2d_s2_tiled_ok(C.byref(d)))


def conv2d_s2_tiled(src: View, w_tiled, dst: View, scale=None, bias=None, act=L.ACT_NONE, res: Optional[View] = None, stats=None,
                    accumulate=False):
    """3x3 stride-2 forward on the tap-resident kernel (csrc/conv_vt.hip); w_tiled = pack_weight_tiled(w)[0]."""
    lib = L.load()
    d = conv_desc(src, dst, 3, 2, L.CONV_FWD, act, res, accumulate)
    L.check(lib.cdet_conv2d_s2_tiled(C.byref(d), ptr(src), ptr(w_tiled), ptr(scale), ptr(bias), ptr(res), ptr(dst), ptr(stats), stream()),
            "cdet_conv2d_s2_tiled")
    return dst


def conv2d_s2_tiled_dgrad(dy: View, w_dgrad_tiled, dx: View, res: Optional[View] = None, accumulate=False):
    """Data gradient of a 3x3 stride-2 convolution: dy [N,Ho,Wo,Cout] -> dx [N,2Ho,2Wo,Cin] (+ res, the gradient already there);
    w_dgrad_tiled = pack_weight_tiled(w, fwd=False, dgrad=True)[1]. dx may be fp32 (accumulate: dx += result)."""
    lib = L.load()
    d = conv_desc(dy, dx, 3, 2, L.CONV_DGRAD, L.ACT_NONE, res, accumulate)
    L.check(lib.cdet_conv2d_s2_tiled_dgrad(C.byref(d), ptr(dy), ptr(w_dgrad_tiled), None, None, ptr(res), ptr(dx), None, stream()),
            "cdet_conv2d_s2_tiled_dgrad")
    return dx


def conv_s2_tiled_stat_blocks(src: View, dst: View) -> int:
    d = conv_desc(src, dst, 3, 2)
    return L.load().cdet_conv2d_s2_tiled_stat_blocks(C.byref(d))


def conv_tiled_stat_blocks(src: View, dst: View, k) -> int:
    d = conv_desc(src, dst, k, 1)
    return L.load().cdet_conv2d_tiled_stat_blocks(C.byref(d))


def conv_stat_blocks(src: View, dst: View, k, s) -> int:
    d = conv_desc(src, dst, k, s)
    return L.load().cdet_conv2d_stat_blocks(C.byref(d))


def conv2d_wgrad(src: View, dy: View, dw: torch.Tensor, k, s, accumulate=False, ws=None):
    """dw: fp32 OIHW [Cd_real, Cs, k, k]; dy may carry channel padding (dy.C >= dw.shape[0])."""
    lib = L.load()
    d = conv_desc(src, dy, k, s)
    d.Cd = dw.shape[0]
    n = lib.cdet_conv2d_wgrad_ws_elems(C.byref(d))
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.float32, device=dw.device)
    L.check(lib.cdet_conv2d_wgrad(C.byref(d), ptr(src), ptr(dy), ptr(dw), ptr(ws), int(accumulate), stream()), "cdet_conv2d_wgrad")
    return dw


def conv2d_wgrad_grouped(items, k, s, accumulate=False):
    """items: [(src View, dy View, dw fp32 OIHW)] of identical geometry -> one launch (cdet_conv2d_wgrad_grouped)."""
    lib = L.load()
    src0, dy0, dw0 = items[0]
    d = conv_desc(src0, dy0, k, s)
    d.Cd = dw0.shape[0]
    d.src_ld = max(it[0].ld for it in items)  # bounds check of the 32-bit DMA offsets uses the widest view
    assert lib.cdet_conv2d_wgrad_groupable(C.byref(d)), "geometry is not taken by the tap-resident weight-gradient kernel"
    tab = (L.WgradItem * len(items))()
    for i, (src, dy, dw) in enumerate(items):
        assert (src.N, src.H, src.W, src.C) == (src0.N, src0.H, src0.W, src0.C) and (dy.ld, dy.coff, dy.C) == (dy0.ld, dy0.coff, dy0.C)
        assert dw.shape == dw0.shape and dw.dtype == torch.float32 and dw.is_contiguous()
        tab[i].x, tab[i].dy, tab[i].dw, tab[i].src_ld, tab[i].src_coff = ptr(src), ptr(dy), dw.data_ptr(), src.ld, src.coff
    tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dw0.device)
    n = lib.cdet_conv2d_wgrad_grouped_ws_elems(C.byref(d), len(items))
    ws = torch.empty(n, dtype=torch.float32, device=dw0.device)
    L.check(lib.cdet_conv2d_wgrad_grouped(C.byref(d), tab_dev.data_ptr(), len(items), ptr(ws), int(accumulate), stream()), "cdet_conv2d_wgrad_grouped")
    return [it[2] for it in items]


def stem_conv(img: torch.Tensor, w: torch.Tensor, dst: View, scale=None, bias=None, act=L.ACT_NONE, stats=None):
    lib = L.load()
    N, c, H, W = img.shape
    assert c == 3 and img.is_contiguous() and dst.coff == 0 and dst.ld == dst.C
    L.check(lib.cdet_stem_conv(ptr(img), dt(img.dtype), ptr(w), ptr(scale), ptr(bias), ptr(dst), N, H, W, dst.C, dt(dst.dtype), act,
                               ptr(stats), stream()), "cdet_stem_conv")
    return dst


def stem_conv1(img: torch.Tensor, w_stem: torch.Tensor, w1: torch.Tensor, dst: View, stem_scale=None, stem_bias=None, scale=None, bias=None,
               act=L.ACT_SILU):
    """Eval-form fusion of the first two backbone rows (csrc/stem_conv1.hip): img NCHW, w_stem fp32 [c1,3,3,3], w1 fp32 [c2,c1,3,3]."""
    lib = L.load()
    N, c, H, W = img.shape
    assert c == 3 and img.is_contiguous()
    c1, c2 = w_stem.shape[0], w1.shape[0]
    ws = torch.empty(lib.cdet_stem_conv1_pack_elems(c1), dtype=dst.dtype, device=img.device)
    L.check(lib.cdet_stem_conv1_pack(ptr(w_stem.detach().float().contiguous()), ptr(ws), c1, dt(dst.dtype), stream()), "cdet_stem_conv1_pack")
    wf, _ = pack_weight_tiled(w1, dst.dtype)
    L.check(lib.cdet_stem_conv1(ptr(img), dt(img.dtype), ptr(ws), ptr(stem_scale), ptr(stem_bias), ptr(wf), ptr(scale), ptr(bias), ptr(dst), N, H, W,
                                c1, c2, dt(dst.dtype), dst.ld, dst.coff, act, stream()), "cdet_stem_conv1")
    return dst


def stem_stat_blocks(N, H, W) -> int:
    return L.load().cdet_stem_conv_stat_blocks(N, H, W)


def stem_conv_wgrad(img, dy: View, dw, accumulate=False):
    lib = L.load()
    N, c, H, W = img.shape
    assert dy.coff == 0 and c == 3 and img.is_contiguous()
    ws = torch.empty(lib.cdet_stem_conv_wgrad_ws_elems(N, H, W), dtype=torch.float32, device=dw.device)
    L.check(lib.cdet_stem_conv_wgrad(ptr(img), dt(img.dtype), ptr(dy), dy.ld, dt(dy.dtype), ptr(dw), N, H, W, dy.C, int(accumulate), ptr(ws),
                                     stream()), "cdet_stem_conv_wgrad")
    return dw


def bn_finalize(stats, nblk, Cn, count, eps, momentum, running_mean, running_var, mean, invstd):
    L.check(L.load().cdet_bn_finalize(ptr(stats), nblk, Cn, count, eps, momentum, ptr(running_mean), ptr(running_var), ptr(mean),
                                      ptr(invstd), stream()), "cdet_bn_finalize")


def bn_silu_fwd(z: View, mean, invstd, gamma, beta, y: View, res: Optional[View] = None):
    L.check(L.load().cdet_bn_silu_fwd(ptr(z), z.ld, z.coff, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(res),
                                      res.ld if res else 0, res.coff if res else 0, ptr(y), y.ld, y.coff, z.M, z.C, dt(z.dtype), stream()),
            "cdet_bn_silu_fwd")
    return y


def bn_bwd_blocks(M) -> int:
    return L.load().cdet_bn_bwd_blocks(M)


def bn_silu_bwd(dy: View, z: View, mean, invstd, gamma, beta, dz: View, dgamma, dbeta, accumulate=False, part=None):
    lib = L.load()
    nblk = lib.cdet_bn_bwd_blocks(z.M)
    need = nblk * 2 * z.C + 2 * z.C
    if part is None or part.numel() < need:
        part = torch.empty(need, dtype=torch.float32, device=z.buf.device)
    L.check(lib.cdet_bn_silu_bwd_reduce(ptr(dy), dy.ld, dy.coff, ptr(z), z.ld, z.coff, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
                                        ptr(part), z.M, z.C, dt(z.dtype), stream()), "cdet_bn_silu_bwd_reduce")
    L.check(lib.cdet_bn_silu_bwd_apply(ptr(dy), dy.ld, dy.coff, ptr(z), z.ld, z.coff, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
                                       ptr(part), nblk, ptr(dgamma), ptr(dbeta), int(accumulate), ptr(dz), dz.ld, dz.coff, z.M, z.C,
                                       dt(z.dtype), 0, stream()), "cdet_bn_silu_bwd_apply")
    return dz


def copy_channels(src: View, dst: View, accumulate=False):
    assert src.C == dst.C and src.M == dst.M
    L.check(L.load().cdet_copy_channels(ptr(src), src.ld, src.coff, ptr(dst), dst.ld, dst.coff, src.M, src.C, dt(src.dtype),
                                        int(accumulate), stream()), "cdet_copy_channels")
    return dst


def add_channels(a: View, b: View, y: View):
    L.check(L.load().cdet_add_channels(ptr(a), a.ld, a.coff, ptr(b), b.ld, b.coff, ptr(y), y.ld, y.coff, a.M, a.C, dt(a.dtype), stream()),
            "cdet_add_channels")
    return y


def upsample2(src: View, dst: View):
    assert dst.H == 2 * src.H and dst.W == 2 * src.W and dst.C == src.C
    L.check(L.load().cdet_upsample2(ptr(src), src.ld, src.coff, ptr(dst), dst.ld, dst.coff, src.N, src.H, src.W, src.C, dt(src.dtype),
                                    stream()), "cdet_upsample2")
    return dst


def upsample2_bwd(ddst: View, dsrc: View, accumulate=False):
    L.check(L.load().cdet_upsample2_bwd(ptr(ddst), ddst.ld, ddst.coff, ptr(dsrc), dsrc.ld, dsrc.coff, dsrc.N, dsrc.H, dsrc.W, dsrc.C,
                                        dt(dsrc.dtype), int(accumulate), stream()), "cdet_upsample2_bwd")
    return dsrc


def sppf_pool(buf: View, Cn):
    """buf: concat buffer; slice [coff, coff+Cn) holds x, slices 1..3 receive the chained 5x5 max pools."""
    L.check(L.load().cdet_sppf_pool(ptr(buf), buf.ld, buf.coff, buf.N, buf.H, buf.W, Cn, dt(buf.dtype), stream()), "cdet_sppf_pool")
    return buf


def sppf_pool_bwd(buf: View, dbuf: View, Cn):
    L.check(L.load().cdet_sppf_pool_bwd(ptr(buf), ptr(dbuf), buf.ld, buf.coff, buf.N, buf.H, buf.W, Cn, dt(buf.dtype), stream()),
            "cdet_sppf_pool_bwd")
    return dbuf


def detect_decode(feats, nc, strides, out_dtype=torch.float32):
    """feats: 3 NHWC tensors [N,h,w,ld>=64+nc] -> y [N, 4+nc, A] (reference layout)."""
    lib = L.load()
    N = feats[0].shape[0]
    ld = feats[0].shape[3]
    assert all(f.is_contiguous() and f.shape[3] == ld for f in feats)
    A = sum(f.shape[1] * f.shape[2] for f in feats)
    y = torch.empty((N, 4 + nc, A), dtype=out_dtype, device=feats[0].device)
    hw = (C.c_int32 * 6)(*[v for f in feats for v in (f.shape[1], f.shape[2])])
    st = (C.c_float * 3)(*[float(s) for s in strides])
    L.check(lib.cdet_detect_decode(ptr(feats[0]), ptr(feats[1]), ptr(feats[2]), hw, st, N, nc, ld, dt(feats[0].dtype), ptr(y), dt(out_dtype),
                                   stream()), "cdet_detect_decode")
    return y


def det_loss(feats, gt, nc, gains, strides, grad_scale=1.0, grad_dtype=None, want_assign=False, topk=10, alpha=0.5, beta=6.0):
    """feats: 3 NHWC tensors [N,h,w,ld]; gt [N,n_max,5] fp32 (cls,x1,y1,x2,y2 px). Returns (loss5, dfeats, assign)."""
    lib = L.load()
    dev = feats[0].device
    N, ld = feats[0].shape[0], feats[0].shape[3]
    d = L.LossDesc()
    d.N, d.nc, d.n_max = N, nc, gt.shape[1]
    for i, f in enumerate(feats):
        d.hw[2 * i], d.hw[2 * i + 1] = f.shape[1], f.shape[2]
        d.stride[i] = float(strides[i])
    d.gain_box, d.gain_cls, d.gain_dfl = gains["box"], gains["cls"], gains["dfl"]
    d.grad_scale = grad_scale
    d.dtype = dt(feats[0].dtype)
    grad_dtype = grad_dtype or feats[0].dtype
    d.grad_dtype = dt(grad_dtype)
    d.f_ld, d.topk, d.alpha, d.beta = ld, topk, alpha, beta
    A = sum(f.shape[1] * f.shape[2] for f in feats)
    ws = torch.empty(lib.cdet_det_loss_ws_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    out = torch.empty(5, dtype=torch.float32, device=dev)
    dfe = [torch.empty(f.shape, dtype=grad_dtype, device=dev) for f in feats]
    asg = None
    if want_assign:
        asg = dict(fg_mask=torch.empty((N, A), dtype=torch.uint8, device=dev), target_gt_idx=torch.empty((N, A), dtype=torch.int32, device=dev),
                   target_labels=torch.empty((N, A), dtype=torch.int32, device=dev), target_bboxes=torch.empty((N, A, 4), device=dev),
                   target_scores=torch.empty((N, A, nc), device=dev))
    g = asg or {}
    L.check(lib.cdet_det_loss(C.byref(d), ptr(feats[0]), ptr(feats[1]), ptr(feats[2]), ptr(gt), ptr(dfe[0]), ptr(dfe[1]), ptr(dfe[2]), ptr(out),
                              ptr(g.get("fg_mask")), ptr(g.get("target_gt_idx")), ptr(g.get("target_labels")), ptr(g.get("target_bboxes")),
                              ptr(g.get("target_scores")), ptr(ws), stream()), "cdet_det_loss")
    return out, dfe, asg


def nms_batched(pred: torch.Tensor, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300,
                max_nms=30000, return_anchor=False):
    """pred [N, 4+nc, A] -> (rows [N,max_det,6] fp32, counts [N] i32[, anchor [N,max_det] i32: where every kept row came from])."""
    lib = L.load()
    assert pred.is_contiguous()
    N, no, A = pred.shape
    nc = no - 4
    d = L.NmsDesc()
    d.N, d.nc, d.A, d.dtype = N, nc, A, dt(pred.dtype)
    d.conf_thres, d.iou_thres = conf_thres, iou_thres
    d.agnostic, d.multi_label, d.max_det, d.max_nms = int(agnostic), int(multi_label), max_det, max_nms
    d.max_cand = A * nc if (multi_label and nc > 1) else A
    cls_t = None
    if classes is not None:
        cls_t = torch.tensor(list(classes), dtype=torch.int32, device=pred.device)
        d.classes, d.n_classes = cls_t.data_ptr(), cls_t.numel()
    ws = torch.empty(lib.cdet_nms_ws_bytes(C.byref(d)), dtype=torch.uint8, device=pred.device)
    rows = torch.zeros((N, max_det, 6), dtype=torch.float32, device=pred.device)
    cnt = torch.zeros(N, dtype=torch.int32, device=pred.device)
    if return_anchor:
        anchor = torch.zeros((N, max_det), dtype=torch.int32, device=pred.device)
        L.check(lib.cdet_nms_batched_idx(C.byref(d), ptr(pred), ptr(rows), ptr(cnt), ptr(anchor), ptr(ws), stream()), "cdet_nms_batched_idx")
        return rows, cnt, anchor
    L.check(lib.cdet_nms_batched(C.byref(d), ptr(pred), ptr(rows), ptr(cnt), ptr(ws), stream()), "cdet_nms_batched")
    return rows, cnt


def merge_tasks(rows_per_task, counts_per_task, cls_offsets, iou_thres=0.8, scale=None):
    """Per-task NMS outputs ([N,max_det,6] fp32 + [N] i32 each, task order) -> (rows [N, T*max_det, 6] with global class ids,
    counts [N]): cross-task suppression, and with scale [N,5] (gain, pad_x, pad_y, h0, w0) scale_boxes().round()."""
    lib = L.load()
    T = len(rows_per_task)
    N, max_det, _ = rows_per_task[0].shape
    d = L.MergeDesc()
    d.N, d.T, d.max_det, d.iou_thres = N, T, max_det, iou_thres
    for t in range(T):
        assert rows_per_task[t].is_contiguous() and rows_per_task[t].dtype == torch.float32 and counts_per_task[t].dtype == torch.int32
        d.rows[t], d.counts[t], d.cls_offset[t] = ptr(rows_per_task[t]), ptr(counts_per_task[t]), int(cls_offsets[t])
    dev = rows_per_task[0].device
    out = torch.zeros((N, T * max_det, 6), dtype=torch.float32, device=dev)
    cnt = torch.zeros(N, dtype=torch.int32, device=dev)
    L.check(lib.cdet_merge_tasks(C.byref(d), ptr(scale), ptr(out), ptr(cnt), stream()), "cdet_merge_tasks")
    return out, cnt


def match_predictions(det_rows: torch.Tensor, det_count: torch.Tensor, labels: torch.Tensor, label_start: torch.Tensor, iouv: torch.Tensor,
                      max_labels: int) -> torch.Tensor:
    """det_rows [N,max_det,6] + det_count [N] i32, labels [L,5] (cls,x1,y1,x2,y2) grouped by image via label_start [N+1] i32,
    iouv [T] -> correct [N,max_det,T] uint8 (val.py:32-54 for the whole batch in one launch)."""
    lib = L.load()
    N, max_det, _ = det_rows.shape
    d = L.MatchDesc()
    d.N, d.max_det, d.T, d.max_labels = N, max_det, iouv.numel(), int(max_labels)
    assert det_rows.is_contiguous() and det_rows.dtype == torch.float32 and det_count.dtype == torch.int32 and label_start.dtype == torch.int32
    labels = labels.contiguous().float()
    if labels.numel() == 0:  # no label in the whole batch: the kernel still wants a valid pointer
        labels = torch.zeros((1, 5), dtype=torch.float32, device=det_rows.device)
    iouv = iouv.contiguous().float()
    out = torch.empty((N, max_det, d.T), dtype=torch.uint8, device=det_rows.device)
    L.check(lib.cdet_match_predictions(C.byref(d), ptr(det_rows), ptr(det_count), ptr(labels), ptr(label_start), ptr(iouv),
                                       ptr(out), stream()), "cdet_match_predictions")
    return out
