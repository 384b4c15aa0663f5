"""Diagnostic for tests/arrangement_check.py: the check's own sequence (in-line warm-up, then the three arrangements), repeated, with
the maximum differences of the forward taps between every pair of runs.  python tools/arr_diag.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from fv2p_harness.fv2p_model import FV2PDetector  # noqa: E402
from test_fv2p_step_gpu import SmallFV2P, make_inputs  # noqa: E402

gpu = torch.device("cuda:0")
torch.manual_seed(3)
model = FV2PDetector(SmallFV2P).to(gpu)
clouds, feats, coords, gt, u = make_inputs(SmallFV2P, 2, 4096)
args = ([c.to(gpu) for c in clouds], feats.to(gpu), coords.to(gpu), gt.to(gpu), u.to(gpu))
model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": False, "point_branch_stream": False})
model(*args).backward()
torch.cuda.synchronize()
keys = ("bev", "prop_scores", "prop_boxes", "point_features", "sampled_rois")
for rep in range(3):
    outs = []
    for dense, point in ((True, True), (False, True), (False, False)):
        model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": dense, "point_branch_stream": point})
        model.taps = {}
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        loss = model(*args)
        loss.backward()
        torch.cuda.synchronize()
        outs.append({k: model.taps[k].clone() for k in keys})
    for i, j in ((0, 1), (0, 2), (1, 2)):
        print("rep", rep, "runs", i, j, {k: float((outs[i][k] - outs[j][k]).abs().max()) for k in keys})
