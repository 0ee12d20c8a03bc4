cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env $1 python bench.py --steps 20 --warmup 5 --no-breakdown --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['north_star_fwd']['ms'], d['inference']['infer_e2e']['pipelined_ms_per_batch'])"; }
run "X=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "X=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "GPU_MAX_HW_QUEUES=16"
run "HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=16"
