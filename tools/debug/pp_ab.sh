#!/bin/bash
# A/B of the ping-pong 3x3 form (csrc/conv_pp.hip) against the 4-wave form on the shapes it takes; tests first.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_tiled.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/pp_tests.txt
SH="--shape 40,40,320,320,3 --shape 80,80,160,160,3 --shape 80,80,320,320,3 --shape 40,40,640,320,3 --shape 80,80,320,160,3"
for r in 1 2; do
for pp in 0 1; do
  echo "== CDET_CONV_PP=$pp (round $r)" >> gpurun_out/pp_ab.txt
  CDET_CONV_PP=$pp timeout 300 python tools/conv_tiled_bench.py $SH 2>&1 | grep -v amdgpu.ids >> gpurun_out/pp_ab.txt
done
done
cat gpurun_out/pp_tests.txt gpurun_out/pp_ab.txt
