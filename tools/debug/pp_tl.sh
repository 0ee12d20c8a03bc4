#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp CDET_LIB_PATH=tools/debug/_build/libcdet_prof.so
O=gpurun_out/pp_tl.txt; : > $O
for s in 40,40,320,320 40,40,640,320; do
  CDET_PP_ABLATE=64 python tools/pp_timeline.py --custom $s --mode silu 2>&1 | grep -v amdgpu.ids >> $O
done
echo "--- bench, profiling build, ABL 0 / 64" >> $O
for a in 0 64; do CDET_PP_ABLATE=$a python tools/conv_tiled_bench.py --rounds 5 --shape 40,40,320,320,3 --shape 40,40,640,320,3 2>&1 | grep -E "^ *[0-9]+x" >> $O; done
cat $O
