#!/bin/bash
# Step time of bench.py (default two-stream schedule) under the tuning switches of the profiling build (tools/build_prof.sh).
cd $GRAFT_REPO_ROOT
export CDET_LIB_PATH=$GRAFT_REPO_ROOT/tools/debug/_build/libcdet_prof.so
run() { printf "%-44s " "$1"; env $1 python bench.py --steps 20 --warmup 5 --no-breakdown --no-cpu-baseline --no-infer 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run "X=1"
run "CDET_BN_BWD_CAP=256"
run "CDET_BN_BWD_CAP=1024"
run "CDET_BN_BWD_DIV=32"
run "CDET_BN_BWD_DIV=128"
run "X=1"
run "CDET_WGRAD_TARGET=448"
run "CDET_WGRAD_TARGET=384"
run "CDET_BN_REV=1"
run "CDET_BN_REV=2"
run "CDET_BN_REV=3"
run "X=1"
