#!/bin/bash
# rocprofv3 kernel stats of the end-to-end inference stream (tools/debug/stream_probe2.py: bench.predict_e2e on the calibrated detector)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
rm -rf gpurun_out/it; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/it -o it -- python3 tools/debug/stream_probe2.py > gpurun_out/it.log 2>&1
S=$(find gpurun_out/it -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
tail -3 gpurun_out/it.log
rm -rf gpurun_out/it
