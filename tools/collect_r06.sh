#!/bin/bash
# Round-6 evidence on the GPU box (via gpurun): bench line, rocprofv3 kernel stats (two-stream and sequential schedules), PMC HBM traffic,
# per-shape forward / convolution tables, forward timeline, dry-comm record. Summaries land in gpurun_out/r06/ (copy to profiles/).
TAG=r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/$TAG
python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -o st -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/stats.log 2>&1
CDET_TASK_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_seq -o st -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/stats_seq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/pmc_write.log 2>&1
S=$(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1); python3 tools/stats_summary.py "$S" 11 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 8 --warmup 3 (default two-stream schedule), round 6" > gpurun_out/$TAG/kernel_stats.txt
S=$(find gpurun_out/$TAG/stats_seq -name "*kernel_stats.csv" | head -1); python3 tools/stats_summary.py "$S" 11 "CDET_TASK_STREAMS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 8 --warmup 3 (sequential schedule: every kernel alone on the GPU), round 6" > gpurun_out/$TAG/kernel_stats_sequential.txt
F=$(dirname $(find gpurun_out/$TAG/pmc_fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find gpurun_out/$TAG/pmc_write -name "*counter_collection.csv" | head -1))
python3 tools/pmc_traffic.py "$F" "$W" gpurun_out/$TAG/pmc_traffic.json > gpurun_out/$TAG/pmc_traffic.log 2>&1
python3 tools/fwd_shapes.py > gpurun_out/$TAG/fwd_shapes_eval.txt 2>&1
python3 tools/conv_shapes.py > gpurun_out/$TAG/conv_shapes.txt 2>&1
rm -rf gpurun_out/ft; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft -o ft -- python3 tools/debug/fwd_trace.py run > gpurun_out/$TAG/ft.log 2>&1; python3 tools/debug/fwd_trace.py parse gpurun_out/ft > gpurun_out/$TAG/fwd_timeline.txt; rm -rf gpurun_out/ft
python3 bench.py --dry-comm > gpurun_out/$TAG/dry_comm.json 2> gpurun_out/$TAG/dry_comm.err
rm -rf gpurun_out/$TAG/stats gpurun_out/$TAG/stats_seq gpurun_out/$TAG/pmc_fetch gpurun_out/$TAG/pmc_write
ls -la gpurun_out/$TAG; tail -c 400 gpurun_out/$TAG/bench.json; head -12 gpurun_out/$TAG/kernel_stats_sequential.txt
