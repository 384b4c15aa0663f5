#!/bin/bash
# One rocprofv3 --pmc pass (counters only, with --kernel-trace) over `tools/microbench.py convone`, averaged per dispatch of the kernel.
#   bash tools/pmc_pass.sh <tag> "<kernel-name substring>" COUNTER [COUNTER ...]      -> gpurun_out/pmc_<tag>.json
TAG=$1; KERNEL=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FV2P_RES=1
rm -rf gpurun_out/pmc_$TAG
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -o pmc -- python3 tools/microbench.py convone > gpurun_out/pmc_$TAG.log 2>&1
python3 - "$TAG" "$KERNEL" <<'PY'
import csv, glob, collections, json, sys
tag, kernel = sys.argv[1], sys.argv[2]
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    if not rows:
        print("no", kernel, "in", f); continue
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"kernel": rows[0]["Kernel_Name"][:90], "dispatches": len(next(iter(acc.values())))}
    for c, v in sorted(acc.items()):
        res[c] = sum(v) / len(v)
    json.dump(res, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
    print(json.dumps(res))
PY
rm -rf gpurun_out/pmc_$TAG
