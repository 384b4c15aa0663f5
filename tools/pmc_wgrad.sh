#!/bin/bash
# MFMA / wait counters of the pair-split weight-gradient kernels (`tools/microbench.py conv`): one --pmc pass with --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d gpurun_out/pmc_wg -o pmc -- python3 tools/microbench.py conv > gpurun_out/pmc_wg.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_wg/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv_wgrad_pairs_dma<1, 1, 32>" in n or "conv_wgrad_pairs_dma<1, 1, 64>" in n or "conv_wgrad_pairs<2, 2>" in n:
            acc[(n.split("(")[0].replace("void fv2p::", ""), r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, g, c), v in sorted(acc.items()):
        print(f"{k:34s} grid {g:>8s} {c:30s} n={len(v):3d} avg {sum(v)/len(v):14.1f}")
PY
