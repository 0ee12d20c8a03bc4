#!/bin/bash
# VERDICT r05 item 4, pricing: what the in-LDS normalisation of the staged pixel tile would add to the consumer convolution (profiling build,
# CDET_HALO_ABLATE=32: one piece = 8 values per lane rewritten per K step) -- against what the stand-alone bn_silu_fwd pass costs.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp CDET_LIB_PATH=tools/debug/_build/libcdet_prof.so CDET_CONV_PP=0
O=gpurun_out/bn_lds_price.txt; : > $O
for r in 1 2; do for a in 0 32; do
  echo "== CDET_HALO_ABLATE=$a (round $r; 4-wave form)" >> $O
  CDET_HALO_ABLATE=$a python tools/conv_tiled_bench.py --rounds 5 --shape 80,80,160,160,3 --shape 40,40,320,320,3 --shape 80,80,320,320,3 2>&1 | grep -E "^ *[0-9]+x" >> $O
done; done
python tools/ew_bench.py 2>&1 | grep -v amdgpu.ids | head -30 >> $O
cat $O
