"""Times the fused first-two-rows launch (csrc/stem_conv1.hip) alone at batch 32 @640, YOLOv8x widths; with the profiling build
(CDET_LIB_PATH=tools/debug/_build/libcdet_prof.so) CDET_SC1_ABLATE selects what is left out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cerberusdet_amd import ops, _lib as L

dev = "cuda"
N = int(os.environ.get("BS", "32"))
dtype = torch.bfloat16
g = torch.Generator().manual_seed(1)
img = torch.rand(N, 3, 640, 640, generator=g).to(dtype).to(dev)
w0 = (torch.randn(80, 3, 3, 3, generator=g) / 5).to(dev)
w1 = (torch.randn(160, 80, 3, 3, generator=g) / 27).to(dev)
s0, b0 = (torch.rand(80, generator=g) + 0.5).to(dev), (torch.randn(80, generator=g) * 0.3).to(dev)
s1, b1 = (torch.rand(160, generator=g) + 0.5).to(dev), (torch.randn(160, generator=g) * 0.3).to(dev)
out = ops.new_act(N, 160, 160, 160, dtype)
lib = L.load()
ws = torch.empty(lib.cdet_stem_conv1_pack_elems(80), dtype=dtype, device=dev)
L.check(lib.cdet_stem_conv1_pack(w0.data_ptr(), ws.data_ptr(), 80, ops.dt(dtype), ops.stream()), "pack")
wf, _ = ops.pack_weight_tiled(w1, dtype)


def run():
    L.check(lib.cdet_stem_conv1(img.data_ptr(), ops.dt(img.dtype), ws.data_ptr(), s0.data_ptr(), b0.data_ptr(), wf.data_ptr(), s1.data_ptr(), b1.data_ptr(),
                                out.buf.data_ptr(), N, 640, 640, 80, 160, ops.dt(dtype), 160, 0, L.ACT_SILU, ops.stream()), "sc1")


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * N * (320 * 320 * 80 * 27 + 160 * 160 * 160 * 720)
print(f"ablate {os.environ.get('CDET_SC1_ABLATE', '0'):>2s}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TF/s", flush=True)
