"""Plain PyTorch fp32 references of the convolution and its two gradients on NHWC tensors, written as shifted GEMMs (torch.matmul).

Test infrastructure only. Why not F.conv2d: on the GPU box it goes through MIOpen (find / JIT on first use of every shape), on the CPU
it does not finish at BASELINE sizes. A 3x3 convolution is nine [M, Cin] x [Cin, Cout] GEMMs over shifted views of the padded
input; with integer-valued operands every product and partial sum is an exact fp32 integer whatever the summation order, so these
references are EXACT there (the full-size identity tests), and ordinary fp32 otherwise (the teacher-forced layer tests).
"""
import torch
import torch.nn.functional as F


def out_hw(H, W, k, s):
    p = k // 2
    return (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1


def conv_fwd(x, w, s=1):
    """x [N,H,W,Ci] fp32, w [Co,Ci,k,k] fp32 (pad k//2) -> [N,Ho,Wo,Co] fp32."""
    N, H, W, Ci = x.shape
    Co, _, k, _ = w.shape
    p = k // 2
    Ho, Wo = out_hw(H, W, k, s)
    xp = F.pad(x, (0, 0, p, p, p, p)) if p else x
    y = torch.zeros((N, Ho, Wo, Co), dtype=torch.float32, device=x.device)
    for kh in range(k):
        for kw in range(k):
            xs = xp[:, kh:kh + s * (Ho - 1) + 1:s, kw:kw + s * (Wo - 1) + 1:s, :]
            y += torch.matmul(xs, w[:, :, kh, kw].t())
    return y


def conv_dgrad(dy, w, s, H, W):
    """dy [N,Ho,Wo,Co] fp32 -> dx [N,H,W,Ci] fp32 (adjoint of conv_fwd)."""
    N, Ho, Wo, Co = dy.shape
    _, Ci, k, _ = w.shape
    p = k // 2
    dxp = torch.zeros((N, H + 2 * p, W + 2 * p, Ci), dtype=torch.float32, device=dy.device)
    for kh in range(k):
        for kw in range(k):
            dxp[:, kh:kh + s * (Ho - 1) + 1:s, kw:kw + s * (Wo - 1) + 1:s, :] += torch.matmul(dy, w[:, :, kh, kw])
    return dxp[:, p:p + H, p:p + W, :].contiguous() if p else dxp


def conv_wgrad(x, dy, k, s=1):
    """x [N,H,W,Ci], dy [N,Ho,Wo,Co] fp32 -> dw [Co,Ci,k,k] fp32."""
    N, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    p = k // 2
    xp = F.pad(x, (0, 0, p, p, p, p)) if p else x
    dw = torch.empty((Co, Ci, k, k), dtype=torch.float32, device=x.device)
    d2 = dy.reshape(-1, Co)
    for kh in range(k):
        for kw in range(k):
            xs = xp[:, kh:kh + s * (Ho - 1) + 1:s, kw:kw + s * (Wo - 1) + 1:s, :].reshape(-1, Ci)
            dw[:, :, kh, kw] = torch.matmul(d2.t(), xs)
    return dw


def silu(a):
    return a * torch.sigmoid(a)


def dsilu(a):
    s_ = torch.sigmoid(a)
    return s_ * (1 + a * (1 - s_))
