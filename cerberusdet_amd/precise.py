"""The fp32-ACCURATE eval forward of a CerberusDet model on the MI355X (round 5) -- `model.full_precision()`.

The reference evaluates a model with fp32 parameters in fp32 (cerberusdet/models/cerberus.py:804-882 on a `.float()` model; `CerberusDetInference`
with half=False, cerberusdet_inference.py:26-40), and BASELINE.json states the tolerance against it: boxes within 1e-3 relative. The compiled
16-bit plans (engine.Plan) store activations in bf16 / fp16 and match that reference statistically. This module is the same
forward at the reference's OWN precision, still on the product's kernels:

  * an fp32 map is carried as its fp32 value plus three bf16 terms, t = hi + mid + lo (exact to 2^-24 |t|); so is every weight;
  * a convolution is the six term pairs whose product lies above 2^-24, accumulated in fp32 by `cdet_conv2d_tiled` / `cdet_conv2d_s2_tiled`
    (conv_halo_kernel / conv_vt_kernel, fp32 destination, accumulate form) -- the 3-channel stem, which those refuse, on `cdet_conv2d`;
  * folded BatchNorm + SiLU + the Bottleneck shortcut (`cdet_epilogue_f32`), the SPPF pools (`cdet_maxpool_f32`), Concat / nn.Upsample
    (`cdet_split3` into a channel slice, optionally through the nearest 2x upsample) run in fp32 and emit value and terms at once; the Detect
    decode is the plans' own fp32 kernel (`cdet_detect_decode`).

No arithmetic on activations happens in torch; torch allocates the buffers. Weights are split by the same `cdet_split3` kernel and packed by
`cdet_pack_weight_tiled` (exact: every term is a bf16 value); the folded BatchNorm scale / bias are computed in float64 on the device.

Train form (model.train()): BatchNorm from the batch statistics of the fp32 accumulator (`cdet_bn_train_f32`, double, fixed order; running statistics
updated), and a BACKWARD on the same footing -- BatchNorm / SiLU backward in fp32 (`cdet_bn_silu_bwd_f32`, emits dz as three terms), the data
gradient as six term-pair launches of the tap-resident kernels on the DGRAD operand / the stride-2 parity-class kernel, the weight gradient as six
launches of `cdet_conv2d_wgrad`, pool / upsample / Concat / shortcut gradients in `cdet_maxpool_bwd_f32` / `cdet_add_f32`, the projections' bias
gradient in `cdet_colsum_f32`; parameter gradients accumulate into the fp32 `.grad` tensors like the 16-bit plans'. `out = model(x, task);
loss(out).backward()` works through the autograd bridge at the bottom of this file. Cost: 6 MFMA launches + 1 elementwise pass per convolution and
fp32 maps -- roughly 8x the time of the bf16 plan; it exists for parity, not for throughput (trainers.Averaging runs it as sequential task passes).

Walk order and graph semantics follow `CerberusDet.execution_plan` / `_inputs` exactly as engine.Plan._build does (reference
cerberus.py:804-882, models/yolo.py:87-100, models/common.py:51-68, 107-117, 174-191, 230-245, 288-295).
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from .models.common import C2f, Concat, Conv, SPPF, Upsample
from .ops import View, conv_desc, detect_decode, dt, pack_weight, pack_weight_tiled, stream

# operand-term pairs whose product is above 2^-24 relative: hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi
_PAIRS = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]


def _up8(c: int) -> int:
    return (c + 7) // 8 * 8


class Map:
    """An fp32 NHWC map [N, H, W, ld] with its three bf16 term buffers of the same geometry; a channel slice [coff, coff + C) of it."""

    __slots__ = ("f", "s", "coff", "C", "ctot", "g")

    def __init__(self, f: Optional[torch.Tensor], s: Optional[List[torch.Tensor]], coff: int, Cn: int, ctot: int, g: Optional[torch.Tensor] = None):
        self.f, self.s, self.coff, self.C, self.ctot = f, s, coff, Cn, ctot  # ctot: real channels of the buffer (ld = ctot rounded up to 8)
        self.g = g  # train plans: d(loss)/d(map), fp32, same geometry (zeroed at the start of every backward; every consumer ACCUMULATES into it)

    @staticmethod
    def new(N, H, W, Cn, dev, terms=True, value=True) -> "Map":
        ld = _up8(Cn)
        f = torch.zeros((N, H, W, ld), dtype=torch.float32, device=dev) if value else None  # (zeros: the pad channels are read by the convolutions)
        s = [torch.zeros((N, H, W, ld), dtype=torch.bfloat16, device=dev) for _ in range(3)] if terms else None
        return Map(f, s, 0, Cn, Cn)

    def _any(self):
        return self.f if self.f is not None else self.s[0]

    N = property(lambda m: m._any().shape[0])
    H = property(lambda m: m._any().shape[1])
    W = property(lambda m: m._any().shape[2])
    ld = property(lambda m: m._any().shape[3])
    M = property(lambda m: m.N * m.H * m.W)

    def slice(self, c0: int, c: int) -> "Map":
        return Map(self.f, self.s, self.coff + c0, c, self.ctot, self.g)

    def term(self, i: int) -> View:
        """Term i as a convolution source: a slice that ends the buffer's real channels is widened over the (zero) pad channels behind it, so
        that its channel count is a multiple of 8 like every interior slice's."""
        return View(self.s[i], self.coff, self.C if self.coff + self.C < self.ctot else self.ld - self.coff)

    def tptr(self, i):
        return self.s[i].data_ptr() if self.s is not None else None

    def fptr(self):
        return self.f.data_ptr() if self.f is not None else None


class PrecisePlan:
    """Launch list of one (tasks, shape) configuration: compiled once, replayed per forward; packed weight terms refreshed when parameters change."""

    def __init__(self, model, tasks: Sequence[str], N: int, H: int, W: int, img_dtype: torch.dtype, device, training: bool = False):
        self.lib = L.load()
        self.model, self.tasks, self.N, self.H, self.W, self.img_dtype, self.device = model, list(tasks), N, H, W, img_dtype, device
        self.training = training        # train-form BatchNorm (batch statistics, running statistics updated); forward only
        self.bns: List[nn.BatchNorm2d] = []
        self._ws: Optional[torch.Tensor] = None
        self._ws_doubles = 0
        self.bwd: List = []             # train plans: one closure per forward unit, executed in reverse
        self._grads: List[torch.Tensor] = []   # every map-gradient buffer (zeroed per backward)
        self._gparams: List[torch.nn.Parameter] = []  # parameters whose .grad the backward accumulates into
        self.dfeats: Dict[str, List[torch.Tensor]] = {}
        self._wg_ws: Optional[torch.Tensor] = None
        self._wg_elems = 0
        self.generation = 0
        self._img_map: Optional[Map] = None
        self.steps: List = []           # closures, in launch order
        self.packs: List = []           # closures that (re)build packed weight terms / folded scale and bias
        self.counts = {"tiled": 0, "s2_tiled": 0, "generic": 0, "epilogue": 0, "split": 0, "pool": 0, "bn": 0}  # launches per forward, by entry point
        self._scratch: Optional[torch.Tensor] = None
        self._scratch_elems = 0
        self._late: List = []
        # frozen parameters (reference models/cerberus.py:885-925 freeze_shared_layers: requires_grad False, no gradient at all): no weight / gamma /
        # beta / bias gradient is computed for them, and a map none of whose producers has a trainable parameter carries no gradient buffer -- the
        # data-gradient chain stops above an all-frozen trunk, as engine.Plan does for the 16-bit plans (`frozen`). The mask is part of the plan's
        # cache key (models/cerberus.py::full_precision_plan).
        self._rg = True
        self.feats: Dict[str, List[torch.Tensor]] = {}
        self.y: Dict[str, torch.Tensor] = {}
        self._img: List[Optional[torch.Tensor]] = [None]
        self._packed_at = None
        self._build()
        self._scratch = torch.empty(self._scratch_elems, dtype=torch.float32, device=device)
        self._ws = torch.empty(max(self._ws_doubles, 1), dtype=torch.float64, device=device)
        self._wg_ws = torch.empty(max(self._wg_elems, 1), dtype=torch.float32, device=device)
        for bind in self._late:
            bind()

    # ------------------------------------------------------------------------------------------------ launches
    def _split(self, src_ptr_fn, src_dtype, src_ld, src_coff, nchw, up, dst: Map, want_value=True):
        lib, args = self.lib, (src_dtype, src_ld, src_coff, int(nchw), int(up))
        N, H, W, Cn = dst.N, dst.H, dst.W, dst.C

        def run():
            L.check(lib.cdet_split3(src_ptr_fn(), *args, dst.fptr() if want_value else None, dst.tptr(0), dst.tptr(1), dst.tptr(2), dst.ld, dst.coff,
                                    N, H, W, Cn, stream()), "cdet_split3")
        self.steps.append(run)
        self.counts["split"] += 1

    def _epilogue(self, rec: "_ConvRec", scale, bias, act, res: Optional[Map], y: Map):
        """y = act(z * scale + bias) + res over the first y.C channels of the convolution's accumulator."""
        lib = self.lib

        def run():
            L.check(lib.cdet_epilogue_f32(rec.z.data_ptr(), rec.Op, 0, scale() if scale else None, bias() if bias else None, act,
                                          res.fptr() if res else None, res.ld if res else 0, res.coff if res else 0, y.fptr(), y.tptr(0), y.tptr(1), y.tptr(2),
                                          y.ld, y.coff, y.M, y.C, stream()), "cdet_epilogue_f32")
        self.steps.append(run)
        self.counts["epilogue"] += 1

    def _conv_raw(self, x: Map, weight: torch.nn.Parameter, k: int, s: int, Ho: int, Wo: int, keep_z: bool = False) -> "_ConvRec":
        """z[:, :Op] = conv(x, weight) in fp32 (six bf16 term-pair launches) into the shared accumulator, or -- keep_z, the train form, whose backward
        reads it again -- into one of its own. The record also carries what the backward of this convolution needs."""
        lib, dev = self.lib, self.device
        O, Ci = weight.shape[0], weight.shape[1]
        assert Ci == x.C, (Ci, x.C)
        Op = _up8(O)
        xv = x.term(0)
        assert xv.C % 8 == 0 and x.coff % 8 == 0, f"full_precision: channel slice [{x.coff}, {x.coff + x.C}) is not a multiple of 8 wide"
        zv = _GeomView(x.N, Ho, Wo, Op, torch.float32)
        rec = _ConvRec()
        rec.x, rec.xC, rec.Op, rec.O, rec.Ci, rec.k, rec.s, rec.Ho, rec.Wo, rec.weight = x, xv.C, Op, O, Ci, k, s, Ho, Wo, weight
        if keep_z:
            rec.z = torch.empty(x.N * Ho * Wo * Op, dtype=torch.float32, device=dev)
        else:
            self._scratch_elems = max(self._scratch_elems, x.N * Ho * Wo * Op)
            self._late.append(lambda: setattr(rec, "z", self._scratch))
        d0 = conv_desc(xv, zv, k, s)
        d0.accumulate = 1
        if s == 1 and lib.cdet_conv2d_tiled_ok(C.byref(d0)):
            path, fn = "tiled", lib.cdet_conv2d_tiled
        elif s == 2 and k == 3 and lib.cdet_conv2d_s2_tiled_ok(C.byref(d0)):
            path, fn = "s2_tiled", lib.cdet_conv2d_s2_tiled
        else:
            path, fn = "generic", lib.cdet_conv2d
        # the data gradient's kernel (train plans; not for the image): dX = conv of dz with the DGRAD operand / the stride-2 parity-class kernel
        need_dx = self.training and x is not self._img_map and x.g is not None
        rec.need_dx, dpath, dfn = need_dx, None, None
        if need_dx:
            dzv, dxv = _GeomView(x.N, Ho, Wo, Op, torch.bfloat16), _GeomView(x.N, x.H, x.W, xv.C, torch.float32, x.ld, x.coff)
            if s == 1 and lib.cdet_conv2d_tiled_ok(C.byref(_acc(conv_desc(dzv, dxv, k, 1)))):
                dpath, dfn, rec.ddesc = "tiled", lib.cdet_conv2d_tiled, _acc(conv_desc(dzv, dxv, k, 1))
            elif s == 2 and k == 3 and lib.cdet_conv2d_s2_tiled_ok(C.byref(conv_desc(dzv, dxv, 3, 2, L.CONV_DGRAD))):
                dpath, dfn, rec.ddesc = "s2_tiled", lib.cdet_conv2d_s2_tiled_dgrad, _acc(conv_desc(dzv, dxv, 3, 2, L.CONV_DGRAD))
            else:
                dpath, dfn, rec.ddesc = "generic", lib.cdet_conv2d, _acc(conv_desc(dzv, dxv, k, s, L.CONV_DGRAD))
        rec.dfn, rec.dpath = dfn, dpath
        rec.want_dw = bool(self.training and weight.requires_grad)
        if rec.want_dw:
            wd = conv_desc(xv, _GeomView(x.N, Ho, Wo, Op, torch.bfloat16), k, s)
            wd.Cd = Op
            rec.wdesc = wd
            rec.dw = torch.zeros((Op, xv.C, k, k), dtype=torch.float32, device=dev)
            self._wg_elems = max(self._wg_elems, int(lib.cdet_conv2d_wgrad_ws_elems(C.byref(wd))))
            self._gparams.append(weight)
        # weight terms: padded fp32 OIHW -> cdet_split3 (as a [1, 1, numel] map) -> three packed operands (and three DGRAD operands)
        wpad = torch.zeros((Op, xv.C, k, k), dtype=torch.float32, device=dev)
        terms = [torch.zeros(wpad.numel(), dtype=torch.bfloat16, device=dev) for _ in range(3)]
        packed: List[Optional[torch.Tensor]] = [None, None, None]
        rec.packed_d = [None, None, None]

        def pack():
            wpad[:O, :Ci].copy_(weight.detach().float())
            n = wpad.numel()
            L.check(lib.cdet_split3(wpad.data_ptr(), L.F32, n, 0, 0, 0, None, terms[0].data_ptr(), terms[1].data_ptr(), terms[2].data_ptr(), n, 0,
                                    1, 1, 1, n, stream()), "cdet_split3")
            for i in range(3):
                w32 = terms[i].float().view_as(wpad)  # (exact widening of bf16 values; the packers take fp32 OIHW)
                packed[i] = pack_weight(w32, torch.bfloat16) if path == "generic" else pack_weight_tiled(w32, torch.bfloat16)[0]
                if need_dx:
                    rec.packed_d[i] = (pack_weight(w32, torch.bfloat16, transpose=True, o_pad=Op) if dpath == "generic"
                                       else pack_weight_tiled(w32, torch.bfloat16, fwd=False, dgrad=True)[1])
        self.packs.append(pack)
        descs = []
        for n_ in range(len(_PAIRS)):
            d = conv_desc(xv, zv, k, s)
            d.accumulate = 1 if n_ > 0 else 0
            descs.append(d)

        def run():
            st, zp = stream(), rec.z.data_ptr()
            for n_, (i, j) in enumerate(_PAIRS):
                L.check(fn(C.byref(descs[n_]), x.s[i].data_ptr(), packed[j].data_ptr(), None, None, None, zp, None, st), "precise convolution")
        self.steps.append(run)
        self.counts[path] += len(_PAIRS)
        return rec

    def _conv_backward(self, rec: "_ConvRec", dz_terms: List[torch.Tensor]):
        """dW (+= into weight.grad) and dX (+= into grad(x)) of one convolution from the three terms of dz [N, Ho, Wo, Op] (pad channels zero)."""
        lib, x = self.lib, rec.x
        st = stream()
        if rec.want_dw:  # (a frozen weight gets no gradient: the reference leaves .grad None on requires_grad=False parameters)
            for n_, (i, j) in enumerate(_PAIRS):   # dW = sum over term pairs x_i (x) dz_j
                L.check(lib.cdet_conv2d_wgrad(C.byref(rec.wdesc), x.s[i].data_ptr(), dz_terms[j].data_ptr(), rec.dw.data_ptr(), self._wg_ws.data_ptr(),
                                              1 if n_ > 0 else 0, st), "precise weight gradient")
            _grad_of(rec.weight).add_(rec.dw[:rec.O, :rec.Ci])
        if rec.need_dx:
            # dX += conv(dz_i, W_j^T): every launch accumulates (other consumers of x may have written already). The three entry points -- tiled kernel on the
            # DGRAD operand, stride-2 parity-class kernel, generic kernel in DGRAD mode -- share one argument list
            for i, j in _PAIRS:
                L.check(rec.dfn(C.byref(rec.ddesc), dz_terms[i].data_ptr(), rec.packed_d[j].data_ptr(), None, None, None, x.g.data_ptr(), None, st),
                        "precise data gradient")

    def _conv_unit(self, m: Conv, x: Map, y: Map, res: Optional[Map] = None):
        """SiLU(BN(conv(x))) (+ res) -> y (models/common.py:51-68; BatchNorm folded as in fuseforward, or the fused module's own bias). Train plans:
        BatchNorm from the batch statistics, and a backward closure (shortcut fan-in, BatchNorm / SiLU backward, weight and data gradient)."""
        Ho, Wo = (x.H + 2 * (m.k // 2) - m.k) // m.s + 1, (x.W + 2 * (m.k // 2) - m.k) // m.s + 1
        assert (Ho, Wo, m.c2) == (y.H, y.W, y.C), ((Ho, Wo, m.c2), (y.H, y.W, y.C))
        rec = self._conv_raw(x, m.conv.weight, m.k, m.s, Ho, Wo, keep_z=self.training)
        Op = rec.Op
        scale = torch.empty(m.c2, dtype=torch.float32, device=self.device)
        bias = torch.empty(m.c2, dtype=torch.float32, device=self.device)
        if self.training:
            # train-form BatchNorm: statistics of THIS batch from the fp32 accumulator (cdet_bn_train_f32), running statistics updated unless the
            # module is frozen (reference cerberus.py:885-905: track_running_stats False, batch statistics still used)
            if getattr(m, "fused", False):
                raise RuntimeError("training needs un-fused Conv modules (BatchNorm present)")
            bn, lib, M = m.bn, self.lib, self.N * Ho * Wo
            self.bns.append(bn)
            self._gparams += [p_ for p_ in (bn.weight, bn.bias) if p_.requires_grad]
            mean = torch.empty(m.c2, dtype=torch.float32, device=self.device)
            invstd = torch.empty(m.c2, dtype=torch.float32, device=self.device)
            self._ws_doubles = max(self._ws_doubles, int(lib.cdet_bn_train_f32_ws_doubles(m.c2)))

            def run():
                upd = bool(bn.track_running_stats and bn.training)
                L.check(lib.cdet_bn_train_f32(rec.z.data_ptr(), Op, 0, M, m.c2, bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(bn.momentum),
                                              bn.running_mean.data_ptr() if upd else None, bn.running_var.data_ptr() if upd else None, self._ws.data_ptr(),
                                              scale.data_ptr(), bias.data_ptr(), mean.data_ptr(), invstd.data_ptr(), stream()), "cdet_bn_train_f32")
                if upd:
                    bn._nbt_pending = getattr(bn, "_nbt_pending", 0) + 1   # num_batches_tracked (host counter, flushed by the model like the plans')
            self.steps.append(run)
            self.counts["bn"] += 1
            dz = [torch.zeros((self.N, Ho, Wo, Op), dtype=torch.bfloat16, device=self.device) for _ in range(3)]  # (pad channels stay zero)

            def backward():
                st = stream()
                if res is not None and res.g is not None:   # y = res + SiLU(...): the shortcut takes grad(y) as it is (models/common.py:107-117)
                    L.check(lib.cdet_add_f32(y.g.data_ptr(), y.ld, y.coff, res.g.data_ptr(), res.ld, res.coff, y.N, y.H, y.W, y.C, 0, 1, st), "cdet_add_f32")
                L.check(lib.cdet_bn_silu_bwd_f32(y.g.data_ptr(), y.ld, y.coff, rec.z.data_ptr(), Op, scale.data_ptr(), bias.data_ptr(), mean.data_ptr(),
                                                 invstd.data_ptr(), M, m.c2, self._ws.data_ptr(),
                                                 _grad_of(bn.weight).data_ptr() if bn.weight.requires_grad else None,
                                                 _grad_of(bn.bias).data_ptr() if bn.bias.requires_grad else None,
                                                 None, dz[0].data_ptr(), dz[1].data_ptr(), dz[2].data_ptr(), Op, st), "cdet_bn_silu_bwd_f32")
                self._conv_backward(rec, dz)
            if y.g is not None:  # (None: neither this layer nor anything above it trains -- the backward never comes here)
                self.bwd.append(backward)
        else:
            def pack():
                if getattr(m, "fused", False):
                    scale.fill_(1.0)
                    bias.copy_(m.conv.bias.detach().float())
                else:
                    bn = m.bn
                    sc = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
                    scale.copy_(sc.float())
                    bias.copy_((bn.bias.detach().double() - bn.running_mean.double() * sc).float())
            self.packs.append(pack)
        self._epilogue(rec, lambda: scale.data_ptr(), lambda: bias.data_ptr(), L.ACT_SILU, res, y)

    # ------------------------------------------------------------------------------------------------ layers
    def _new(self, H, W, Cn, terms=True, value=True) -> Map:
        m = Map.new(self.N, H, W, Cn, self.device, terms, value)
        if self.training and self._rg:  # (no trainable parameter at or above this layer: nothing would ever read the gradient)
            m.g = torch.zeros((self.N, H, W, m.ld), dtype=torch.float32, device=self.device)
            self._grads.append(m.g)
        return m

    def _add_grad(self, src: Map, dst: Map, down=False):
        """backward of _copy_into: grad(dst map's source) += grad(slice), through the 2x2 down-sum for an upsampled copy. src: the slice that was written."""
        lib = self.lib

        def backward():
            L.check(lib.cdet_add_f32(src.g.data_ptr(), src.ld, src.coff, dst.g.data_ptr(), dst.ld, dst.coff, dst.N, dst.H, dst.W, dst.C, int(down), 1, stream()),
                    "cdet_add_f32")
        if src.g is not None and dst.g is not None:
            self.bwd.append(backward)

    def _copy_into(self, src: Map, dst: Map, up=False):
        """dst slice <- src (value and terms), optionally through the nearest 2x upsample."""
        assert dst.C == src.C
        self._split(lambda: src.f.data_ptr(), L.F32, src.ld, src.coff, False, up, dst, want_value=dst.f is not None)
        if self.training:
            self._add_grad(dst, src, down=up)

    def _layer(self, m, xs: List[Map]) -> Map:
        # does anything at or above this layer train? (inputs that carry a gradient buffer, or a trainable parameter of the layer itself)
        self._rg = any(getattr(v.src if isinstance(v, _Up) else v, "g", None) is not None for v in xs) or any(p_.requires_grad for p_ in m.parameters())
        try:
            return self._layer_body(m, xs)
        finally:
            self._rg = True

    def _layer_body(self, m, xs: List[Map]) -> Map:
        if isinstance(m, Conv):
            x = xs[0]
            y = self._new((x.H + 2 * (m.k // 2) - m.k) // m.s + 1, (x.W + 2 * (m.k // 2) - m.k) // m.s + 1, m.c2)
            self._conv_unit(m, x, y)
            return y
        if isinstance(m, C2f):
            x, c = xs[0], m.c
            cat = self._new(x.H, x.W, (2 + len(m.m)) * c)
            self._conv_unit(m.cv1, x, cat.slice(0, 2 * c))
            cur = cat.slice(c, c)
            for i, b in enumerate(m.m):
                t = self._new(x.H, x.W, b.cv1.c2)
                self._conv_unit(b.cv1, cur, t)
                dst = cat.slice((2 + i) * c, c)
                self._conv_unit(b.cv2, t, dst, res=cur if b.add else None)
                cur = dst
            y = self._new(x.H, x.W, m.cv2.c2)
            self._conv_unit(m.cv2, cat, y)
            return y
        if isinstance(m, SPPF):
            x, c_ = xs[0], m.cv1.c2
            cat = self._new(x.H, x.W, 4 * c_)
            self._conv_unit(m.cv1, x, cat.slice(0, c_))
            lib = self.lib
            for i in range(3):
                a, b = cat.slice(i * c_, c_), cat.slice((i + 1) * c_, c_)

                def run(a=a, b=b):
                    L.check(lib.cdet_maxpool_f32(a.fptr(), a.ld, a.coff, b.fptr(), b.tptr(0), b.tptr(1), b.tptr(2), b.ld, b.coff, b.N, b.H, b.W, b.C, m.k,
                                                 stream()), "cdet_maxpool_f32")
                self.steps.append(run)
                self.counts["pool"] += 1
                if self.training and a.g is not None:
                    def pool_bwd(a=a, b=b):
                        L.check(lib.cdet_maxpool_bwd_f32(a.fptr(), a.ld, a.coff, b.g.data_ptr(), b.ld, b.coff, a.g.data_ptr(), a.ld, a.coff, a.N, a.H, a.W, a.C,
                                                         m.k, stream()), "cdet_maxpool_bwd_f32")
                    self.bwd.append(pool_bwd)
            y = self._new(x.H, x.W, m.cv2.c2)
            self._conv_unit(m.cv2, cat, y)
            return y
        if isinstance(m, Upsample):
            return _Up(xs[0])
        if isinstance(m, Concat):
            srcs = [(v.src, True) if isinstance(v, _Up) else (v, False) for v in xs]
            H, W = (srcs[0][0].H * 2, srcs[0][0].W * 2) if srcs[0][1] else (srcs[0][0].H, srcs[0][0].W)
            cat = self._new(H, W, sum(v.C for v, _ in srcs))
            off = 0
            for v, up in srcs:
                self._copy_into(v, cat.slice(off, v.C), up)
                off += v.C
            return cat
        raise NotImplementedError(f"full_precision: layer {type(m).__name__}")

    def _real(self, v) -> Map:
        if isinstance(v, _Up):  # an Upsample that is not consumed by a Concat
            y = self._new(v.src.H * 2, v.src.W * 2, v.src.C)
            self._copy_into(v.src, y, up=True)
            return y
        return v

    def _detect(self, head, task: str, xs: List[Map]):
        nc = head.nc
        ncp = _up8(nc)
        feats, dfeats, lib = [], [], self.lib
        for lvl, xl in enumerate(xs):
            xl = self._real(xl)
            fb = Map(torch.zeros((self.N, xl.H, xl.W, 64 + ncp), dtype=torch.float32, device=self.device), None, 0, 64 + nc, 64 + nc)
            df = torch.zeros_like(fb.f) if self.training else None   # d(loss)/d(head map): filled by the autograd bridge
            for br, off, cn in ((head.cv2[lvl], 0, 64), (head.cv3[lvl], 64, nc)):
                t1 = self._new(xl.H, xl.W, br[0].c2)
                self._conv_unit(br[0], xl, t1)
                t2 = self._new(xl.H, xl.W, br[1].c2)
                self._conv_unit(br[1], t1, t2)
                proj: nn.Conv2d = br[2]
                rec = self._conv_raw(t2, proj.weight, 1, 1, xl.H, xl.W)
                pb = torch.empty(cn, dtype=torch.float32, device=self.device)
                self.packs.append(lambda pb=pb, proj=proj: pb.copy_(proj.bias.detach().float()))
                self._epilogue(rec, None, lambda pb=pb: pb.data_ptr(), L.ACT_NONE, None, fb.slice(off, cn))
                if self.training:
                    if proj.bias.requires_grad:
                        self._gparams.append(proj.bias)
                    dz = [torch.zeros((self.N, xl.H, xl.W, rec.Op), dtype=torch.bfloat16, device=self.device) for _ in range(3)]
                    M = self.N * xl.H * xl.W

                    def backward(rec=rec, dz=dz, df=df, off=off, cn=cn, proj=proj, M=M, H=xl.H, W=xl.W):
                        st = stream()
                        # the projection has no BatchNorm: dz IS the head map's gradient (its channel slice), split into terms; db = its column sums
                        L.check(lib.cdet_split3(df.data_ptr(), L.F32, df.shape[3], off, 0, 0, None, dz[0].data_ptr(), dz[1].data_ptr(), dz[2].data_ptr(), rec.Op, 0,
                                                self.N, H, W, cn, st), "cdet_split3")
                        if proj.bias.requires_grad:
                            L.check(lib.cdet_colsum_f32(df.data_ptr(), df.shape[3], off, M, cn, self._ws.data_ptr(), _grad_of(proj.bias).data_ptr(), st), "cdet_colsum_f32")
                        self._conv_backward(rec, dz)
                    self.bwd.append(backward)
            feats.append(fb.f)
            dfeats.append(df)
        self.feats[task] = feats
        if self.training:
            self.dfeats[task] = dfeats
        strides = [float(s) for s in head.stride]

        def run():
            self.y[task] = detect_decode(feats, nc, strides)
        if not self.training:  # (train mode returns the raw maps only, models/yolo.py:87-89)
            self.steps.append(run)

    # ------------------------------------------------------------------------------------------------ graph walk
    def _build(self):
        model = self.model
        order, _ = model.execution_plan(self.tasks)
        img = Map.new(self.N, self.H, self.W, 3, self.device)  # three image channels in an 8-channel row (pad channels stay zero); no gradient
        self._img_map = img
        self._split(lambda: self._img[0].data_ptr(), dt(self.img_dtype), 0, 0, True, False, img)
        outs: Dict[int, object] = {}
        for idx in order:
            blk = model.blocks[idx]
            if idx == 0:
                ys, cur = [], img
                for li, lay in enumerate(blk.model):
                    f = lay.f
                    xin = [cur] if (li == 0 or f == -1) else ([ys[f]] if isinstance(f, int) else [cur if j == -1 else ys[j] for j in f])
                    if not isinstance(lay, Concat):
                        xin = [self._real(v) for v in xin]
                    cur = self._layer(lay, xin)
                    ys.append(cur)
                outs[0] = ys
                continue
            xs = [outs[0][j] if kind == "bb" else outs[j] for kind, j in model._inputs[idx]]
            if idx in model.heads.values():
                self._detect(blk, model.controllers[idx].task_id, xs)
            else:
                outs[idx] = self._layer(blk, xs if isinstance(blk, Concat) else [self._real(v) for v in xs])

    # ------------------------------------------------------------------------------------------------ run
    def _versions(self):
        # (train form: the running statistics are written by every forward and read by none of the packs)
        return (self.model._weights_version, sum(int(p._version) for p in self.model.parameters()),
                0 if self.training else sum(int(b._version) for b in self.model.buffers()))

    def release(self):
        """(plan-cache eviction hook, as engine.Plan.release: nothing is registered on the modules here)"""
        self.steps, self.packs = [], []

    def run(self, img: torch.Tensor):
        assert tuple(img.shape) == (self.N, 3, self.H, self.W) and img.dtype == self.img_dtype and img.is_contiguous()
        ver = self._versions()
        if ver != self._packed_at:
            with torch.no_grad():
                for p in self.packs:
                    p()
            self._packed_at = ver
        self._img[0] = img
        for step in self.steps:
            step()
        self._img[0] = None
        self.generation += 1

    def run_backward(self):
        """Backward of the last train-form forward from self.dfeats (d loss / d head maps, filled by the caller): parameter gradients are ACCUMULATED into
        the model-owned fp32 `.grad` tensors, like the 16-bit plans and the reference's per-task passes (trainers/averaging.py:142-168)."""
        assert self.training
        with torch.no_grad():
            for g in self._grads:
                g.zero_()
            for p in self._gparams:
                _grad_of(p)
            for fn in reversed(self.bwd):
                fn()


class _Up:
    """nn.Upsample(None, 2, 'nearest') of a map, not materialised until something other than a Concat needs it."""

    def __init__(self, src: Map):
        self.src = src


class _GeomView:
    """Geometry of a map for conv_desc (the pointer is bound at launch)."""

    def __init__(self, N, H, W, Cn, dtype, ld=None, coff=0):
        self.N, self.H, self.W, self.C, self.ld, self.coff, self.dtype = N, H, W, Cn, (Cn if ld is None else ld), coff, dtype


class _ConvRec:
    """What one convolution of the plan leaves for its epilogue and its backward."""

    __slots__ = ("x", "xC", "Op", "O", "Ci", "k", "s", "Ho", "Wo", "weight", "z", "need_dx", "dfn", "dpath", "ddesc", "wdesc", "dw", "packed_d", "want_dw")


def _acc(d):
    d.accumulate = 1
    return d


def _grad_of(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, dtype=torch.float32)
    return p.grad


class _PreciseFunction(torch.autograd.Function):
    """autograd bridge of a train-form full-precision plan (the counterpart of autograd_bridge._PlanFunction for the 16-bit plans)."""

    @staticmethod
    def forward(ctx, plan, x, anchor):
        plan.run(x)
        ctx.plan, ctx.generation = plan, plan.generation
        outs = []
        for t in plan.tasks:
            nc = plan.model.get_head(t).nc
            outs += [f.clone()[..., :64 + nc].permute(0, 3, 1, 2) for f in plan.feats[t]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("backward() of a forward whose plan has run forward again since: the plan keeps ONE set of saved accumulators per "
                               "configuration -- call backward() before the next forward of the same configuration")
        i = 0
        for t in plan.tasks:
            nc = plan.model.get_head(t).nc
            for d in plan.dfeats[t]:
                g = grads[i]
                i += 1
                d.zero_()
                if g is not None:
                    d[..., :64 + nc].copy_(g.permute(0, 2, 3, 1))
        plan.run_backward()
        return None, None, None


def run_with_autograd(plan: PrecisePlan, x: torch.Tensor):
    anchor = plan.model._autograd_anchor()
    outs = _PreciseFunction.apply(plan, x, anchor)
    res, i = {}, 0
    for t in plan.tasks:
        res[t] = list(outs[i:i + 3])
        i += 3
    return res
