#!/bin/bash
# SQ / LDS / TA counter passes over one shape of tools/conv_tiled_bench.py (both kernels run; rows are per kernel name).
# Usage: tools/pmc_halo.sh <shape index> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IDX=${1:-0}; TAG=${2:-pmc_halo}
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_LEVEL_LDS" \
            "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/${TAG}_$i -- python3 tools/conv_tiled_bench.py --only $IDX --rounds 1 --reps 3 > gpurun_out/${TAG}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/${TAG}_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'conv_igemm' in k or 'conv_halo' in k:
                a = acc[(k.split('(')[0][-50:], r['Counter_Name'])]
                a[0] += float(r['Counter_Value']); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{k:50s} {c:34s} per-launch {v / n:16.1f}  (n={n})")
PY
