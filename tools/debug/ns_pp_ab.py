import os, sys, json, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
model, cfg = bench.build_model('v8x_2task.yaml', dev)
for dt in (torch.bfloat16, torch.float16):
    r = bench.north_star_forward(model, dev, bs=32, imgsz=640, dtype=dt)
    print(os.environ.get('CDET_CONV_PP', 'default'), 'bf16' if dt == torch.bfloat16 else 'fp16', r['ms'], r['frac'])
