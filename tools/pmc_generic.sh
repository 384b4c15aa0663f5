#!/bin/bash
# One rocprofv3 --pmc pass (counters + --kernel-trace only) over any command, averaged per dispatch and kernel-name substring.
#   bash tools/pmc_generic.sh <out.json> "<kernel substring>[;<kernel substring>...]" "COUNTER COUNTER ..." -- <program> [args]
OUT=$1; KERNELS=$2; COUNTERS=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/pmc_tmp_$$
rm -rf $D
rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d $D -o pmc -- "$@" > $D.log 2>&1
python3 - "$OUT" "$KERNELS" "$D" "$*" <<'PY'
import csv, glob, collections, json, sys
out, kernels, d, cmd = sys.argv[1:5]
res = {"command": cmd, "kernels": {}}
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for kernel in kernels.split(";"):
        sel = [r for r in rows if kernel in r["Kernel_Name"]]
        if not sel:
            continue
        acc = collections.defaultdict(list)
        for r in sel:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        e = {"name": sel[0]["Kernel_Name"][:100], "dispatches": len(next(iter(acc.values())))}
        for c, v in sorted(acc.items()):
            e[c] = sum(v) / len(v)
        res["kernels"][kernel] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $D $D.log
