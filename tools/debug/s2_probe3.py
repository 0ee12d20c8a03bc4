import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from cerberusdet_amd import ops, _lib as L
DEV = "cuda"
N, H, W, Ci, Co, dtype = 1, 40, 40, 32, 160, torch.bfloat16
g = torch.Generator().manual_seed(3)
x = torch.randint(-2, 3, (N, Ci, H, W), generator=g).float()
w = (torch.rand(Co, Ci, 3, 3, generator=g) < 0.3).float() * (torch.randint(0, 2, (Co, Ci, 3, 3), generator=g) * 2 - 1).float()
ref = F.conv2d(x, w, None, 2, 1)
src = ops.from_nchw(x.to(DEV), dtype)
wf, _ = ops.pack_weight_tiled(w.to(DEV), dtype)
for rep in range(3):
    dst = ops.new_act(N, H // 2, W // 2, Co, dtype)
    ops.conv2d_s2_tiled(src, wf, dst)
    torch.cuda.synchronize()
    got = dst.nchw().float().cpu()
    bad = got != ref
    print("rep", rep, int(bad.sum()), "bad; nan", int(torch.isnan(got).sum()), "per-frag", [int(bad[:, f*32:(f+1)*32].sum()) for f in range(5)])
