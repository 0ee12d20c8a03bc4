#!/bin/bash
# Collects the round's evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC HBM traffic passes (separate runs).
# Usage (via gpurun): bash tools/collect_profiles.sh r01f
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -o st -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/stats.log 2>&1
CDET_TASK_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_seq -o st -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/stats_seq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-breakdown > gpurun_out/$TAG/pmc_write.log 2>&1
rm -f gpurun_out/$TAG/stats*/st_kernel_trace.csv gpurun_out/$TAG/pmc_*/pmc_kernel_trace.csv
find gpurun_out/$TAG -type f | head -30
tail -c 600 gpurun_out/$TAG/bench.json
