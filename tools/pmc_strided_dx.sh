#!/bin/bash
# MFMA counters of the strided backward-data conv in plain row order (conv_rows_dma<.., true>) and on parity-ordered
# tiles (conv_rows_act<.., true>): both are launched by `tools/microbench.py conv`.  One --pmc pass with --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 \
  --kernel-trace --output-format csv -d gpurun_out/pmc_dx -o pmc -- python3 tools/microbench.py conv > gpurun_out/pmc_dx.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_dx/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if ("conv_rows_act<64, 2, true>" in n or "conv_rows_dma<64, 2, true>" in n or "conv_rows_act<64, 4, true>" in n) and r["Grid_Size"] in ("203264", "117888"):
            acc[(n.split("(")[0].replace("void fv2p::", ""), r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, g, c), v in sorted(acc.items()):
        print(f"{k:34s} grid {g:>7s} {c:30s} n={len(v):3d} avg {sum(v)/len(v):14.1f}")
PY
