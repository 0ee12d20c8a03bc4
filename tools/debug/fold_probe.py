"""One-off: where do the folded and the two-kernel BatchNorm finalize differ (running statistics)?"""
import ctypes as C
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from cerberusdet_amd import _lib as L
from cerberusdet_amd.ops import conv_desc
import test_gpu_bn_fold as T

lib = L.load()
name, N, H, W, Ci, Co, k, s = T.GEOMS[0]
xs, wt, z = T._conv_case(N, H, W, Ci, Co, k, s, seed=len(name))
d = conv_desc(xs[0], z, k, s)
nblk = lib.cdet_conv2d_tiled_stat_blocks(C.byref(d))
M = z.M
st = torch.cuda.current_stream().cuda_stream
fold = T.Fold()
f32 = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device="cuda")
stats_a, stats_b = f32(nblk * 2 * Co), f32(nblk * 2 * Co)
mean_a, inv_a, mean_b, inv_b, totals = f32(Co), f32(Co), f32(Co), f32(Co), f32(2 * Co)
rm_a, rv_a, rm_b, rv_b = f32(Co), torch.ones(Co, device="cuda"), f32(Co), torch.ones(Co, device="cuda")
fd = fold.desc(nblk, Co, totals=totals, mean=mean_b, invstd=inv_b, running_mean=rm_b, running_var=rv_b, inv_count=1.0 / M, unbias=M / (M - 1), eps=T.EPS, momentum=T.MOM)
for it in range(3):
    x = xs[it % 3]
    ra0, va0, rb0, vb0 = rm_a.clone(), rv_a.clone(), rm_b.clone(), rv_b.clone()
    L.check(lib.cdet_conv2d_tiled(C.byref(d), x.buf.data_ptr(), wt.data_ptr(), None, None, None, z.buf.data_ptr(), stats_a.data_ptr(), st), "conv")
    L.check(lib.cdet_bn_finalize(stats_a.data_ptr(), nblk, Co, M, T.EPS, T.MOM, rm_a.data_ptr(), rv_a.data_ptr(), mean_a.data_ptr(), inv_a.data_ptr(), st), "fin")
    L.check(lib.cdet_conv2d_tiled_bn(C.byref(d), x.buf.data_ptr(), wt.data_ptr(), z.buf.data_ptr(), stats_b.data_ptr(), fd.data_ptr(), st), "conv_bn")
    torch.cuda.synchronize()
    keep = torch.tensor(1.0, dtype=torch.float32) - torch.tensor(T.MOM, dtype=torch.float32)
    want_rm = (keep.cuda() * ra0) + (torch.tensor(T.MOM, dtype=torch.float32).cuda() * mean_a)
    print(it, "rm diff", float((rm_a - rm_b).abs().max()), "rv diff", float((rv_a - rv_b).abs().max()), "in equal", bool(torch.equal(ra0, rb0)),
          "a vs torch", float((rm_a - want_rm).abs().max()), "b vs torch", float((rm_b - want_rm).abs().max()), "tickets", int(fold.tickets.abs().sum()))
