#!/bin/bash
# Experiments build of the library (-DCDET_EXPERIMENTS: the opt-in forms kept for the record -- 384- / 512-pixel convolution tiles, in-launch BatchNorm fold) into
# tools/debug/_build_exp/libcdet_exp.so (git-ignored; select it with CDET_LIB_PATH; tests/test_gpu_bn_fold.py and the ng = 3 / 4 cases of tests/test_gpu_conv_tiled.py run on it only). The product library is untouched.
set -e
cd "$(dirname "$0")/../cerberusdet_amd/csrc"
OUT=../../tools/debug/_build_exp
mkdir -p $OUT/obj
# the source list is the Makefile's (one place to add a file)
SRCS=$(sed -n 's/^SRCS *= *//p' Makefile)
for s in $SRCS; do
  f=${s%.hip}
  if [ ! -f $OUT/obj/$f.o ] || [ $f.hip -nt $OUT/obj/$f.o ] || [ common.h -nt $OUT/obj/$f.o ] || [ switches.h -nt $OUT/obj/$f.o ] || [ halo_common.h -nt $OUT/obj/$f.o ] || [ wgrad_tr.h -nt $OUT/obj/$f.o ] || [ bn_fold.h -nt $OUT/obj/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-lambda-capture -DCDET_EXPERIMENTS $EXTRA -c $f.hip -o $OUT/obj/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj/*.o -o $OUT/libcdet_exp.so
echo built $OUT/libcdet_exp.so
