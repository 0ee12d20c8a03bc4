import math, sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from cerberusdet_amd import ops, _lib as L
DEV = "cuda"
def run(N, H, W, Ci, Co, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(21)
    x = torch.randint(-2, 3, (N, Ci, H, W), generator=g).float()
    w0 = (torch.rand(Co, Ci, 3, 3, generator=g) < 0.3).float() * (torch.randint(0, 2, (Co, Ci, 3, 3), generator=g) * 2 - 1).float()
    src = ops.from_nchw(x.to(DEV), dtype)
    print(f"fwd N{N} {H}x{W} {Ci}->{Co}")
    for ky in range(3):
        for kx in range(3):
            w = torch.zeros_like(w0); w[:, :, ky, kx] = w0[:, :, ky, kx]
            ref = F.conv2d(x, w, None, 2, 1)
            wf, wd = ops.pack_weight_tiled(w.to(DEV), dtype, fwd=True, dgrad=True)
            Ho, Wo = ref.shape[2:]
            dst = ops.new_act(N, Ho, Wo, Co, dtype)
            ops.conv2d_s2_tiled(src, wf, dst)
            torch.cuda.synchronize()
            got = dst.nchw().float().cpu()
            bad = (got != ref)
            msg = ""
            if bad.any():
                # is the wrong output another tap's result?
                for ky2 in range(3):
                    for kx2 in range(3):
                        w2 = torch.zeros_like(w0); w2[:, :, ky2, kx2] = w0[:, :, ky, kx]
                        if torch.equal(F.conv2d(x, w2, None, 2, 1)[bad], got[bad]): msg += f" [bad outputs = these weights applied at tap ({ky2},{kx2})]"
                zero = float((got[bad] == 0).float().mean())
                msg += f" zero-frac {zero:.2f}; per-frag {[int(bad[:, f*32:(f+1)*32].sum()) for f in range((Co+31)//32)]}; rows(y) with errors {sorted(set(bad.nonzero()[:,2].tolist()))[:8]} cols {sorted(set(bad.nonzero()[:,3].tolist()))[:8]}"
            print(f"   only tap ({ky},{kx}): {int(bad.sum())}/{bad.numel()} differ{msg}")
for args in [(1, 40, 40, 32, 160)]:
    run(*args)
