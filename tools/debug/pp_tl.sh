#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp CDET_LIB_PATH=tools/debug/_build/libcdet_prof.so
O=gpurun_out/pp_tl.txt; : > $O
for s in 40,40,320,320 40,40,640,320 80,80,160,160 80,80,320,320; do
  python tools/pp_timeline.py --custom $s --mode silu 2>&1 | grep -v amdgpu.ids >> $O
done
python tools/pp_timeline.py --custom 40,40,320,320 --mode raw 2>&1 | grep -v amdgpu.ids >> $O
echo "--- with per-phase stamps" >> $O
CDET_PP_ABLATE=32 python tools/pp_timeline.py --custom 40,40,320,320 2>&1 | grep -v amdgpu.ids >> $O
cat $O
