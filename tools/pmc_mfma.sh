#!/bin/bash
# MFMA / wait-state counters of the roofline conv kernel (MI355X_MICROARCH.md, SQ block: 8 slots per pass).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o pmc -- python3 tools/microbench.py convone > gpurun_out/pmc_sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_sq/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "conv_rows" in r["Kernel_Name"]]
    if not rows:
        print("no conv rows in", f); continue
    big = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(list)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:42s} {c:32s} n={len(v):3d} avg {sum(v)/len(v):16.1f}")
PY
tail -3 gpurun_out/pmc_sq.log
