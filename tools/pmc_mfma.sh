#!/bin/bash
# MFMA / wait-state counters of the roofline conv kernel (MI355X_MICROARCH.md, SQ block: 8 slots per pass).
#   bash tools/pmc_mfma.sh [round tag, default r02] [kernel-name substring, default "conv_rows_ksplit<128, false"]
# Counters only together with --kernel-trace; writes gpurun_out/<tag>_pmc_mfma.json.
TAG=${1:-r02}
KERNEL=${2:-"conv_rows_ksplit<128, false, 64"}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FV2P_RES=1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o pmc -- python3 tools/microbench.py convone > gpurun_out/pmc_sq.log 2>&1
grep "roofline kernel" gpurun_out/pmc_sq.log
python3 - "$TAG" "$KERNEL" <<'PY'
import csv, glob, collections, json, sys
tag, kernel = sys.argv[1], sys.argv[2]
layer = next((l.strip() for l in open("gpurun_out/pmc_sq.log") if l.startswith("roofline kernel:")), None)
for f in glob.glob("gpurun_out/pmc_sq/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    if not rows:
        print("no", kernel, "in", f); continue
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"kernel": rows[0]["Kernel_Name"][:80], "layer_line": layer, "dispatches": len(next(iter(acc.values()))),
           "command": "bash tools/pmc_mfma.sh (one rocprofv3 --pmc pass of 8 SQ counters with --kernel-trace only; FV2P_RES=1 python3 tools/microbench.py convone)"}
    for c, v in sorted(acc.items()):
        res[c] = sum(v) / len(v)
    # 16x16x4 fp32 MFMA = 2048 flop = 4 MOPS of 512 flop, 32 busy clocks each on one of 1024 SIMDs
    mfma = res["SQ_INSTS_VALU_MFMA_MOPS_F32"] / 4
    res["derived"] = {"mfma_instructions": mfma, "mfma_flops_executed": mfma * 2048,
                      "mfma_busy_clocks_per_simd": res["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024,
                      "note": "SQ_WAVE_CYCLES / SQ_WAIT_* count in units of 4 clocks; busy fraction = mfma_busy_clocks_per_simd / (kernel duration x shader clock)"}
    json.dump(res, open(f"gpurun_out/{tag}_pmc_mfma.json", "w"), indent=1)
    print(json.dumps(res))
PY
