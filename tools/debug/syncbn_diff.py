"""Debug: per-parameter gradient agreement between local-BN runs (twice) and the SyncBN launch list with a 1-rank RCCL group."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import torch, torch.distributed as dist
import synth
from util import load_golden
from test_gpu_trainer import _model, DEV
from cerberusdet_amd.trainers import Averaging

arrays, meta = load_golden("trainer")
_, mmeta = load_golden("model_tiny2")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
res = []
for sync in (False, False, True, True):
    m = _model(meta, mmeta)
    tr = Averaging(torch.device(DEV), m, meta["hyp"], meta["tasks"], epochs=100, nb=1000, use_ema=False, sync_bn=sync)
    t = meta["tasks"][0]
    img = torch.from_numpy(synth.det_image(300, 4, 128)).to(DEV)
    b = synth.make_batch(4, 3, meta["nc"][0], 400)
    out = tr.forward_backward(t, dict(img=img, **{k: torch.from_numpy(v).to(DEV) for k, v in b.items()}), active_tasks=[t])
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    res.append((out.clone(), {k: p.grad.clone() for k, p in named.items() if p.grad is not None}))
    print("loss", out.tolist())
def cmp(i, j):
    worst = []
    for k in res[i][1]:
        a, b = res[i][1][k].flatten().double().cpu(), res[j][1][k].flatten().double().cpu()
        if float(a.norm()) < 1e-9: continue
        cos = float(a @ b / (a.norm() * b.norm()))
        worst.append((cos, float(b.norm() / a.norm()), k))
    worst.sort()
    print(f"--- {i} vs {j}: bit-identical keys {sum(torch.equal(res[i][1][k], res[j][1][k]) for k in res[i][1])}/{len(res[i][1])}")
    for w in worst[:8]: print("   ", w)
cmp(0, 1); cmp(2, 3); cmp(0, 2)
dist.destroy_process_group()
