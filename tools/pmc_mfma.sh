#!/bin/bash
# MFMA utilisation evidence for the dominant kernel instantiations: rocprofv3 counter passes over tools/mfma_util_run.py,
# random vs all-zero operands, the program directly behind `--`. Writes gpurun_out/<tag>.txt (copy it to profiles/).
# Usage (via gpurun): bash tools/pmc_mfma.sh r04_mfma_util
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-mfma_util}
for Z in rand zeros; do
  ZF=""; [ $Z = zeros ] && ZF="--zeros"
  i=0
  for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rm -rf gpurun_out/${TAG}_${Z}_$i
    rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/${TAG}_${Z}_$i -o p -- python3 tools/mfma_util_run.py $ZF > gpurun_out/${TAG}_${Z}_$i.log 2>&1
  done
done
python3 - "$TAG" > gpurun_out/$TAG.txt <<'PY'
import csv, glob, re, sys, collections
tag = sys.argv[1]
print("# MFMA utilisation of the dominant kernels (batch 32 @640 shapes of tools/mfma_util_run.py), rocprofv3 --pmc + --kernel-trace, per launch averages over the")
print("# timed launches (the last 6 dispatches of each kernel). SQ_VALU_MFMA_BUSY_CYCLES counts MFMA-pipe busy cycles summed over all SIMDs (32 per")
print("# v_mfma_f32_32x32x16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs: busy % = BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs); effective clock = GUI_ACTIVE / 8 / wall time.")
print("# SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES are quad-cycle wave sums. rand = random normal operands, zeros = all-zero operands (same instruction stream).")
print(f"{'operands':8s} {'kernel':44s} {'wall us':>8s} {'GUI_ACTIVE':>11s} {'clock GHz':>9s} {'MFMA_BUSY':>13s} {'busy %':>7s} {'WAIT_LDS/WAVE_CYC':>18s} {'MOPS_BF16':>12s}")
for z in ("rand", "zeros"):
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    wall = collections.defaultdict(list)
    for i in (1, 2):
        d = f"gpurun_out/{tag}_{z}_{i}"
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"cdet::([A-Za-z0-9_]+<[^>]*>|[A-Za-z0-9_]+)", r["Kernel_Name"])
                if not m or not re.match(r"conv_halo|conv_pp|conv_pair|conv_vt|wgrad_halo", m.group(1)):
                    continue
                ctr[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if i == 1:
            for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    m = re.search(r"cdet::([A-Za-z0-9_]+<[^>]*>|[A-Za-z0-9_]+)", r["Kernel_Name"])
                    if m and re.match(r"conv_halo|conv_pp|conv_pair|conv_vt|wgrad_halo", m.group(1)):
                        wall[m.group(1)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in ctr:
        c = {n: sum(v[-6:]) / len(v[-6:]) for n, v in ctr[k].items()}
        w = sum(wall[k][-6:]) / max(len(wall[k][-6:]), 1) / 1e3
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        wl = c.get("SQ_WAIT_INST_LDS", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        print(f"{z:8s} {k[:44]:44s} {w:8.1f} {gui:11.0f} {gui / 8 / max(w, 1e-9) / 1e3:9.2f} {busy:13.0f} {100 * busy / max(gui / 8 * 1024, 1):7.1f} {wl:18.3f} {c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0):12.0f}")
PY
cat gpurun_out/$TAG.txt
rm -rf gpurun_out/${TAG}_rand_* gpurun_out/${TAG}_zeros_*
