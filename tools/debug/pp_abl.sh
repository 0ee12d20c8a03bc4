#!/bin/bash
# ablations of the ping-pong kernel (profiling build): what the loop costs without its DMA (1), fragment reads (2), MFMAs (4)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp CDET_LIB_PATH=tools/debug/_build/libcdet_prof.so
O=gpurun_out/pp_abl.txt; : > $O
for abl in 0 1 2 3 4 5 6 7; do
  echo "== CDET_PP_ABLATE=$abl" >> $O
  CDET_PP_ABLATE=$abl python tools/conv_tiled_bench.py --rounds 5 --shape 40,40,320,320,3 --shape 40,40,640,320,3 2>&1 | grep -E "^ *[0-9]+x" >> $O
done
echo "== timeline (CDET_PP_ABLATE=32)" >> $O
CDET_PP_ABLATE=32 python tools/pp_timeline.py --custom 40,40,320,320 2>&1 | grep -v amdgpu.ids >> $O
cat $O
