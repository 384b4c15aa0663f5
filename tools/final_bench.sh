#!/bin/bash
# The four bench lines of a round, one after the other (stdout JSON -> gpurun_out/bench_<workload>.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for w in fv2p mgaf fv2p-waymo backbone; do
  timeout 900 python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  python3 - "$w" <<'PY'
import json, sys
w = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/bench_{w}.json").read().strip().splitlines()[-1])
    print(w, d["ms_per_step"], d["value"], d.get("inline_ms_per_step"), d.get("boundary_ms_per_step"), d.get("vs_restated_structure"), d["roofline"]["frac"])
except Exception as e:
    print(w, "FAILED", e)
PY
done
