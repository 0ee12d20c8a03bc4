"""median cosine between the bf16 plan's and the full-precision plan's gradients of one YOLOv8x task pass, by batch / size (16-bit noise vs a wiring error)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
for bs, sz in ((2, 256), (8, 320), (16, 640)):
    model, _ = bench.build_model("v8x_2task.yaml", torch.device("cuda", 0))
    model.train()
    t = list(model.heads)[0]
    x = torch.rand(bs, 3, sz, sz, generator=torch.Generator().manual_seed(6)).cuda()
    cots = None
    def one_pass():
        global cots
        model.zero_grad(set_to_none=True)
        maps = model(x if model.compute_dtype == torch.float32 else x.bfloat16(), t)
        if cots is None:
            g = torch.Generator().manual_seed(7)
            cots = [torch.randn(f.shape, generator=g).cuda() for f in maps]
        sum((f.float() * c).sum() for f, c in zip(maps, cots)).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
    g16 = one_pass()
    g16b = one_pass()
    model.full_precision()
    g32 = one_pass()
    cos = sorted(float(torch.nn.functional.cosine_similarity(g32[k].flatten(), g16[k].flatten(), dim=0)) for k in g32 if k.endswith("conv.weight"))
    same = all(torch.equal(g16[k], g16b[k]) for k in g16)
    print(f"bs {bs} @{sz}: median cos {cos[len(cos)//2]:.3f} p10 {cos[len(cos)//10]:.3f} worst {cos[0]:.3f} (bf16 plan deterministic: {same})", flush=True)
    cots = None
    del model
    torch.cuda.empty_cache()
