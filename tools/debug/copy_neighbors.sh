cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
rm -rf gpurun_out/cn; CDET_TASK_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cn -o cn -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/cn.log 2>&1
F=$(find gpurun_out/cn -name "*kernel_trace.csv" | head -1)
python3 tools/debug/copy_neighbors.py $F | head -40
rm -rf gpurun_out/cn
