"""Building blocks with the reference's names, constructor arguments and state-dict keys
(ai-forever/CerberusDet cerberusdet/models/common.py:51-68 Conv, 107-117 Bottleneck, 174-191 C2f, 230-245 SPPF,
288-295 Concat), so that model YAMLs and checkpoints are drop-in.

These modules only OWN parameters and describe structure. They are never executed op-by-op through torch: a
`CerberusDet` compiles the block DAG into a static launch list of gfx950 kernels (cerberusdet_amd/engine.py).
Calling a leaf module directly raises -- there is deliberately no eager / CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

BN_EPS = 1e-3       # reference utils/torch_utils.py:184
BN_MOMENTUM = 0.03  # reference utils/torch_utils.py:185


def autopad(k, p=None, d=1):
    """'same' padding (reference models/common.py:42-48)."""
    if d > 1:
        k = d * (k - 1) + 1
    return k // 2 if p is None else p


class _NoEager(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(
            f"{type(self).__name__} is a structural module of cerberusdet_amd: run it through CerberusDet.forward "
            "(compiled gfx950 launch list). There is no eager or CPU path.")


class Conv(_NoEager):
    """SiLU(BatchNorm2d(Conv2d(bias=False)))  -- or SiLU(Conv2d(bias=True)) after fuse()."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1 or act is not True:
            raise NotImplementedError("cerberusdet_amd.Conv supports groups=1, dilation=1, SiLU (all shipped configs)")
        if k not in (1, 3) or s not in (1, 2) or autopad(k, p, d) != k // 2:
            raise NotImplementedError(f"Conv k={k} s={s} p={p}: only k in (1,3), s in (1,2), 'same' padding")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=BN_EPS, momentum=BN_MOMENTUM)
        self.act = nn.SiLU()
        self.c1, self.c2, self.k, self.s = c1, c2, k, s

    @property
    def fused(self) -> bool:
        return not hasattr(self, "bn")

    def fuse_(self):
        """Fold BN (running stats) into the conv (reference utils/torch_utils.py:191-217, models/yolo.py:218-227)."""
        if self.fused:
            return self
        with torch.no_grad():
            bn, conv = self.bn, self.conv
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
            fused = fused.requires_grad_(False).to(conv.weight.device)
            fused.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
            fused.bias.copy_(bn.bias - bn.weight * bn.running_mean / torch.sqrt(bn.running_var + bn.eps))
        self.conv = fused
        del self.bn
        return self


class Bottleneck(_NoEager):
    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0][0] if isinstance(k[0], (tuple, list)) else k[0], 1)
        self.cv2 = Conv(c_, c2, k[1][0] if isinstance(k[1], (tuple, list)) else k[1], 1, g=g)
        self.add = shortcut and c1 == c2


class C2f(_NoEager):
    """cv1 1x1 -> chunk(2) -> n chained Bottlenecks -> concat of (2+n) chunks -> cv2 1x1."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))
        self.c1, self.c2, self.n = c1, c2, n


class SPPF(_NoEager):
    """cv1 1x1 -> three chained MaxPool2d(5,1,2) -> concat(4) -> cv2 1x1."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        if k != 5:
            raise NotImplementedError("SPPF: only k=5")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.c1, self.c2, self.k = c1, c2, k


class Concat(_NoEager):
    def __init__(self, dimension=1):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError("Concat: only the channel dimension")
        self.d = dimension


class Upsample(_NoEager):
    """nn.Upsample(None, 2, 'nearest') of the model YAMLs (resolved for the name `nn.Upsample`)."""

    def __init__(self, size=None, scale_factor=2, mode="nearest"):
        super().__init__()
        if size is not None or scale_factor != 2 or mode != "nearest":
            raise NotImplementedError("Upsample: only (None, 2, 'nearest')")
        self.scale_factor, self.mode = scale_factor, mode
