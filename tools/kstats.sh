#!/bin/bash
# Per-kernel average durations of one conv shape (rocprofv3 --kernel-trace --stats). Usage: tools/kstats.sh <what> <shape> <tag> [reps]
WHAT=$1; SHAPE=$2; TAG=$3; REPS=${4:-50}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ks_$TAG
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$TAG -o ks -- python3 tools/conv_shapes.py --what $WHAT --only $SHAPE --reps $REPS > gpurun_out/ks_$TAG.log 2>&1
python3 - "$TAG" "$REPS" <<'PY'
import csv, glob, sys
from collections import OrderedDict
tag, reps = sys.argv[1], int(sys.argv[2])
fs = glob.glob(f"gpurun_out/ks_{tag}/**/*kernel_trace.csv", recursive=True)
if not fs:
    print("no trace file"); sys.exit(0)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
# the timed loop is the tail of the trace: `reps` launches of the op = the last reps dispatches of each of its kernels
tail = rows[-4 * reps:]
per = OrderedDict()
for r in tail:
    per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in per.items():
    v = v[-reps:]
    print(f"{k[:100]:100s} n {len(v):4d} avg {sum(v) / len(v) / 1e3:8.1f} us")
PY
rm -rf gpurun_out/ks_$TAG
