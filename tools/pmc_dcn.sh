#!/bin/bash
# Counters of the DCN kernels at one layer shape (default: the MGAF head's feature adaption), three separate rocprofv3 --pmc passes with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ block: 8 slots).
#   bash tools/pmc_dcn.sh <out.json> [B C H W dg]
OUT=${1:-gpurun_out/r04_pmc_dcn.json}; shift
SHAPE=${*:-4 256 200 176 4}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
K="dcn_fwd_k;dcn_bwd_col_k;dcn_bwd_weight_k;dcn_col2im_k;dcn_index_k"
B=tools/ubench/dcn_bench
make -s -C tools/ubench dcn_bench || exit 1   # from source against the library of this tree: never a stale binary
bash tools/pmc_generic.sh $OUT.sq.json "$K" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" -- $B $SHAPE 3 > /dev/null
bash tools/pmc_generic.sh $OUT.fetch.json "$K" "FETCH_SIZE" -- $B $SHAPE 3 > /dev/null
bash tools/pmc_generic.sh $OUT.write.json "$K" "WRITE_SIZE" -- $B $SHAPE 3 > /dev/null
python3 - "$OUT" "$SHAPE" <<'PY'
import json, sys
out, shape = sys.argv[1], [int(v) for v in sys.argv[2].split()]
b, c, h, w, dg = shape
sq, fe, wr = (json.load(open(f"{out}.{k}.json")) for k in ("sq", "fetch", "write"))
npix = b * h * w
flops = 2.0 * npix * c * c * 9
res = {"command": "bash tools/pmc_dcn.sh (three rocprofv3 --pmc passes with --kernel-trace only over tools/ubench/dcn_bench %s 3)" % " ".join(map(str, shape)),
       "layer": f"DCNv2 [{b},{c}->{c},{h},{w}] dg={dg}", "alg_flops_forward": flops,
       "notes": "FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts requests of 128 B and more at half their bytes (MI355X_MICROARCH.md, HBM), but 64-byte "
                "segments - the unit these kernels gather x in: 16 channels of one pixel per four lanes - at their full size (tools/ubench/fetch_calib.hip, "
                "profiles/r04_fetch_calib.json: streaming 16 B/lane 0.50, 512-B and 256-B rows in random order 0.52, 64-B segments in random order 1.00): "
                "fetch_bytes_raw is therefore the estimate for the gather-dominated kernels (the coalesced offset / mask / dy reads inside it are under-counted by "
                "up to half of their share), fetch_bytes_x2 an upper bound.  SQ_WAVE_CYCLES / SQ_WAIT_* count in units of 4 clocks; 16x16x4 fp32 MFMA = 4 MOPS, "
                "32 busy clocks on one of 1024 SIMDs.",
       "kernels": {}}
for k, e in sq["kernels"].items():
    r = dict(e)
    mf = e.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / 4
    r["mfma_instructions"], r["mfma_flops_executed"] = mf, mf * 2048
    r["mfma_busy_clocks_per_simd"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024
    f = fe["kernels"].get(k, {}).get("FETCH_SIZE")
    wv = wr["kernels"].get(k, {}).get("WRITE_SIZE")
    if f is not None:
        r["fetch_bytes_raw"], r["fetch_bytes_x2"] = f * 1024, f * 2048
    if wv is not None:
        r["write_bytes"] = wv * 1024
    res["kernels"][k] = r
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -f $OUT.sq.json $OUT.fetch.json $OUT.write.json
