#!/bin/bash
# PMC passes over one conv shape (rocprofv3 --pmc only with --kernel-trace). Usage: tools/pmc_conv.sh <what> <shape> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WHAT=${1:-fwd}; SHAPE=${2:-40x40-320-320-3}; TAG=${3:-pmc}
i=0
for CTRS in "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES" \
            "TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL" \
            "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCR_TCP_STALL_CYCLES" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/${TAG}_$i -- python3 tools/conv_shapes.py --what $WHAT --only $SHAPE --reps 3 > gpurun_out/${TAG}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/${TAG}_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'conv_igemm' in k or 'conv_wgrad' in k:
                a = acc[(k.split('(')[0][-60:], r['Counter_Name'])]
                a[0] += float(r['Counter_Value']); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{k:60s} {c:36s} per-launch {v / n:16.1f}  (n={n})")
PY
