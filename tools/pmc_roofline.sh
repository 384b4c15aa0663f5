#!/bin/bash
# HBM traffic of the roofline kernel (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE in separate --pmc passes,
# counters only together with --kernel-trace.  Run on the GPU box:  bash tools/pmc_roofline.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 tools/microbench.py convone > gpurun_out/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "conv_rows_dma<64, 4, false>" in r["Kernel_Name"] and r["Counter_Name"] == c]
        big = max(int(r["Grid_Size"]) for r in rows) if rows else 0
        rows = [r for r in rows if int(r["Grid_Size"]) == big]
        vals = [float(r["Counter_Value"]) for r in rows]
        if vals:
            print(c, "kernel", rows[0]["Kernel_Name"][:60], "dispatches", len(vals), "avg counter (KiB)", sum(vals) / len(vals))
PY
