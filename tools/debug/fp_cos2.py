"""per-parameter cosine (execution order) between the bf16 plan's and the full-precision plan's gradients, and the rel-L2 of the train-mode maps"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
bs, sz = 8, 320
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
    return [f.detach().float().clone() for f in maps], {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
m16, g16 = one_pass()
model.half()
xb = x
def one_pass_h():
    model.zero_grad(set_to_none=True)
    maps = model(x.half(), t)
    sum((f.float() * c).sum() for f, c in zip(maps, cots)).backward()
    torch.cuda.synchronize()
    return [f.detach().float().clone() for f in maps], {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
mh, gh = one_pass_h()
model.full_precision()
m32, g32 = one_pass()
for i, (a, b) in enumerate(zip(m16, m32)):
    print(f"map {i}: rel-L2 {float((a - b).norm() / b.norm()):.4f}")
for i, (a, b) in enumerate(zip(mh, m32)):
    print(f"fp16 map {i}: rel-L2 {float((a - b).norm() / b.norm()):.4f}")
order, _ = model.execution_plan([t])
for idx in order:
    ks = [k for k in g32 if k.startswith(f"blocks.{idx}.") and (k.endswith("conv.weight") or k.endswith(".2.weight"))]
    cs = [float(torch.nn.functional.cosine_similarity(g32[k].flatten(), gh[k].flatten(), dim=0)) for k in ks]
    if cs:
        print(f"fp16 block {idx:2d}: {len(cs):2d} conv weights, cos min {min(cs):.3f} med {sorted(cs)[len(cs)//2]:.3f} max {max(cs):.3f}")
for idx in order:
    ks = [k for k in g32 if k.startswith(f"blocks.{idx}.") and (k.endswith("conv.weight") or k.endswith(".2.weight"))]
    cs = [float(torch.nn.functional.cosine_similarity(g32[k].flatten(), g16[k].flatten(), dim=0)) for k in ks]
    rl = [float(g16[k].norm() / (g32[k].norm() + 1e-30)) for k in ks]
    if cs:
        print(f"block {idx:2d}: {len(cs):2d} conv weights, cos min {min(cs):.3f} med {sorted(cs)[len(cs)//2]:.3f} max {max(cs):.3f}; norm ratio 16/32 {min(rl):.2f}..{max(rl):.2f}")
