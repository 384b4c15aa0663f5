"""The contraction audit (tools/fma_audit.py) at reduced size: the oracle compiled with fused multiply-adds (what nvcc does to the
reference's kernels by default) against the plain build every parity test uses.  Not a parity test — a measurement of how far the
"bit-exact vs a non-contracted restatement" claim is from "bit-exact vs a CUDA build": the full-size numbers are committed as
profiles/r03_fma_audit.json and quoted in DESIGN.md 4."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


@pytest.mark.skipif("fma" not in open("/proc/cpuinfo").read(), reason="host CPU without FMA")
def test_contracted_oracle_runs_and_stays_close():
    import fma_audit
    r = fma_audit.audit(points=2048, clouds=1, nms_boxes=1500)
    assert r["fps"]["rounds"] == 2048
    # a contraction changes distances in the last bit: picks may swap at exact-tie-like rounds, never wholesale
    assert r["fps"]["picks_that_differ_per_cloud"][0] <= 2048 // 10
    assert r["three_nn"]["rows_with_another_neighbour"] <= r["three_nn"]["rows_compared"] // 100
    assert r["ball_query"]["centres_with_another_member_list"] <= r["ball_query"]["centres"] // 100
    assert r["points_in_boxes"]["points_with_another_box"] <= 2
    for v in r["nms"].values():
        assert v["in_one_list_only"] <= max(2, v["survivors"] // 100)
    assert r["bev_iou_512x512"]["max_abs_difference"] < 1e-4
