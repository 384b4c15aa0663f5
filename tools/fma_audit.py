#!/usr/bin/env python3
"""Floating-point contraction audit (CPU only).

The reference's setup.py files pass no -fmad=false (pcdet/ops/*/setup.py, setup.py:38-47), so nvcc fuses every product that feeds a
sum into one fused multiply-add: `(x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)` (sampling_gpu.cu:139, interpolate_gpu.cu:37-55,
ball_query_gpu.cu:38-40), the cross products and rotations of iou3d_nms_kernel.cu:36-225 and the point-in-box rotation of
roiaware_pool3d_kernel.cu:27-37.  This repo's kernels and its oracle evaluate those expressions WITHOUT contraction
(-ffp-contract=off on both sides), so "bit-exact against the oracle" is bit-exact against a non-contracted restatement.  This tool
measures what the choice changes: the oracle's C compiled twice (oracle/Makefile: liboracle.so without, liboracle_fma.so with
contraction — gcc's -ffp-contract=fast, which fuses the same expression shape; which of two products in a*b + c*d is the fused one
may differ from nvcc's choice) on the BASELINE-shaped inputs, integer outputs compared one by one.

    python tools/fma_audit.py [--points 16384] [--out profiles/r03_fma_audit.json]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "from-voxel-to-point_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def audit(points=16384, keypoints=None, clouds=2, nms_boxes=9000):
    import oracle
    from fv2p_harness import synth
    keypoints = keypoints or points
    res = {"points_per_cloud": points, "clouds": clouds}
    fps_diff, fps_first, nn_idx_diff, nn_rows, bq_diff, bq_rows, pib_diff, pib_pts = [], [], 0, 0, 0, 0, 0, 0
    for seed in range(clouds):
        pts, boxes = synth.lidar_cloud(seed, points, return_boxes=True)
        xyz = pts[None, :, :3].copy()
        v, c, k = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
        centres = ((c[:, ::-1].astype(np.float32) + 0.5) * synth.KITTI_VOXEL + synth.KITTI_RANGE[:3]).astype(np.float32)

        kp_plain = xyz[0][oracle.furthest_point_sample(xyz, keypoints)[0][0]]   # the queries of both 3-NN runs

        def run():
            keys, _ = oracle.furthest_point_sample(xyz, keypoints)
            d, i = oracle.three_nn_batch(kp_plain[None], centres[None])
            # RoI-head ball query shapes: 512 points around each box, 216 grid centres, radii 0.8 / 1.6
            bq = []
            for bx in boxes[:8]:
                near = np.argsort(((xyz[0, :, :2] - bx[:2]) ** 2).sum(1))[:512]
                g = np.stack(np.meshgrid(*[np.linspace(-0.5, 0.5, 6)] * 3, indexing="ij"), -1).reshape(-1, 3) * bx[3:6] + bx[:3]
                for r, ns in ((0.8, 16), (1.6, 32)):
                    bq.append(oracle.ball_query_batch(r, ns, xyz[:, near], g[None].astype(np.float32)))
            pib = oracle.points_in_boxes_gpu(xyz, boxes[None, :, :7].astype(np.float32))
            return keys[0], i[0], bq, pib[0]
        plain = run()
        with oracle.contracted():
            fused = run()
        same = plain[0] == fused[0]
        fps_diff.append(int((~same).sum()))
        fps_first.append(int(np.argmin(same)) if not same.all() else -1)
        nn_idx_diff += int((plain[1] != fused[1]).any(1).sum())
        nn_rows += plain[1].shape[0]
        for a, b in zip(plain[2], fused[2]):
            bq_diff += int((a != b).any(-1).sum())
            bq_rows += a.shape[0] * a.shape[1]
        pib_diff += int((plain[3] != fused[3]).sum())
        pib_pts += plain[3].size
    res["fps"] = {"picks_that_differ_per_cloud": fps_diff, "first_differing_round_per_cloud": fps_first, "rounds": keypoints}
    res["three_nn"] = {"rows_with_another_neighbour": nn_idx_diff, "rows_compared": nn_rows}
    res["ball_query"] = {"centres_with_another_member_list": bq_diff, "centres": bq_rows}
    res["points_in_boxes"] = {"points_with_another_box": pib_diff, "points": pib_pts}
    nms = {}
    for name, bx, thr in (("proposal_like_9000_thr0.8", synth.proposal_boxes(1, nms_boxes, tight=True), 0.8),
                          ("spread_9000_thr0.8", synth.proposal_boxes(1, nms_boxes), 0.8),
                          ("proposal_like_4096_thr0.1", synth.proposal_boxes(2, min(4096, nms_boxes), tight=True), 0.1)):
        order = -np.arange(bx.shape[0], dtype=np.float32)
        a = oracle.nms(bx, order, thr)
        with oracle.contracted():
            b = oracle.nms(bx, order, thr)
        nms[name] = {"survivors": int(len(a)), "survivors_contracted": int(len(b)), "in_one_list_only": int(len(set(a.tolist()) ^ set(b.tolist())))}
    res["nms"] = nms
    q = synth.proposal_boxes(3, 512)
    iou = oracle.boxes_bev(q, q, "iou")
    with oracle.contracted():
        iou_f = oracle.boxes_bev(q, q, "iou")
    res["bev_iou_512x512"] = {"max_abs_difference": float(np.abs(iou - iou_f).max()), "pairs_that_differ": int((iou != iou_f).sum()), "pairs": int(iou.size)}
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--clouds", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r03_fma_audit.json"))
    a = ap.parse_args()
    r = audit(a.points, clouds=a.clouds)
    json.dump(r, open(a.out, "w"), indent=1)
    print(json.dumps(r, indent=1))
