import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from cerberusdet_amd import ops, _lib as L
DEV = "cuda"
N, H, W, Ci, Co, dtype = 1, 40, 40, 32, 160, torch.bfloat16
x = torch.ones(N, Ci, H, W)
src = ops.from_nchw(x.to(DEV), dtype)
for (ky, kx, k) in [(0, 0, 0), (0, 2, 0), (0, 2, 9), (1, 1, 0), (2, 2, 5)]:
    w = torch.zeros(Co, Ci, 3, 3)
    w[:, k, ky, kx] = torch.arange(1, Co + 1).float()
    ref = F.conv2d(x, w, None, 2, 1)
    wf, _ = ops.pack_weight_tiled(w.to(DEV), dtype)
    dst = ops.new_act(N, H // 2, W // 2, Co, dtype)
    ops.conv2d_s2_tiled(src, wf, dst)
    torch.cuda.synchronize()
    got = dst.nchw().float().cpu()
    bad = (got != ref) | torch.isnan(got)
    print(f"tap ({ky},{kx}) k={k}: {int(bad.sum())} bad, {int(torch.isnan(got).sum())} nan")
    if bad.any():
        badpix = bad.any(1)[0]
        ys, xs = badpix.nonzero()[:, 0], badpix.nonzero()[:, 1]
        print("   bad pixel count", int(badpix.sum()), "first", [(int(a), int(b)) for a, b in zip(ys[:6], xs[:6])])
        y0, x0 = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
        print(f"   pixel ({y0},{x0}) got by cout:", [(c, float(got[0, c, y0, x0])) for c in range(0, 160, 7)])
        print("   bad couts at that pixel:", [c for c in range(160) if bad[0, c, y0, x0]][:50])
