"""Child process of tests/test_fv2p_step_gpu.py::test_stream_arrangements_give_the_same_step (not collected by pytest)."""
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
# the first step of a process runs on the calling stream only, as in bench.py: MIOpen's first-call solver search on a side stream is the
# one thing that was ever seen to hang the dense-branch arrangement
model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": False, "point_branch_stream": False})
model(*args).backward()
torch.cuda.synchronize()
runs = []
compared = 0
for dense, point, wgrad in ((True, True, False), (False, True, False), (False, False, False), (True, True, True)):
    model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": dense, "point_branch_stream": point, "dense_wgrad_stream": wgrad})
    model.taps = {}
    model.zero_grad(set_to_none=True)
    torch.manual_seed(11)   # the RoI head's dropout masks: the same in every arrangement
    loss = model(*args)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((loss.item(), model.taps["keypoints"].clone(), model.taps["sampled_rois"].clone(),
                 {k: p.grad.clone() for k, p in model.named_parameters()},
                 (float(model.taps["loss_rpn"]), float(model.taps["loss_point"])), model.taps["prop_scores"].clone()))
names = ("dense branch on a side stream", "point branch on a side stream", "one stream",
         "dense branch on a side stream, its weight gradients on the weight-gradient stream (bench.py's arrangement)")
for name, other in zip(names[1:], runs[1:]):
    assert torch.equal(other[1], runs[0][1]), f"{name}: other key points than with the {names[0]}"
    # the first stage and the point head do not depend on which RoIs the second stage samples: always compared
    for what, a, b in zip(("anchor-head loss", "point-head loss"), other[4], runs[0][4]):
        assert abs(a - b) < 1e-5 * max(1.0, abs(b)), f"{name}: {what} {a} against {b}"
    # The BEV map is not bit-reproducible from one forward pass to the next in every process (1e-5 relative between IDENTICAL runs
    # in some sequences, bit-identical in others: tools/arr_diag.py; the sparse levels and the decoder are bit-identical), so the
    # proposals move by ~1e-6 m: "the same RoIs" is a tolerance, and a swapped pair of NMS neighbours ends the comparison.
    if other[2].shape != runs[0][2].shape or float((other[2] - runs[0][2]).abs().max()) > 1e-3:
        print(f"{name}: other sampled RoIs ({int(((other[2] - runs[0][2]).abs() > 1e-3).any(-1).sum())} rows; proposal scores differ by at most "
              f"{float((other[5] - runs[0][5]).abs().max()):.2e}) - second stage not compared")
        continue
    compared += 1
    assert abs(other[0] - runs[0][0]) < 1e-4 * max(1.0, abs(runs[0][0])), f"{name}: loss {other[0]} against {runs[0][0]}"
    for k, g0 in runs[0][3].items():
        # In a process whose BEV map is not reproducible the 1e-5 of the forward pass grows on the way back: the sparse backbone and
        # the decoder sit at the far end of the backward chain (the tolerances of test_fv2p_step_gpu.py: 2e-3, 2e-2 at the deep end).
        # (The bias of a conv that feeds BatchNorm has a zero gradient up to rounding — 4e-6 in norm was seen in a process with the
        # non-reproducible map: absolute floor beside the relative bound.)
        rel = 2e-2 if k.startswith(("backbone_3d.", "post_pfe.")) else 2e-3
        err, bound = float((other[3][k] - g0).norm()), rel * float(g0.norm()) + 2e-5 * g0.numel() ** 0.5
        assert err < bound, f"{name}: gradient of {k} differs by {err:.3e} (bound {bound:.3e})"
assert compared >= 1, "no arrangement sampled the same RoIs as the first: nothing of the second stage was compared"
print(f"ARRANGEMENTS AGREE (second stage compared in {compared} of {len(runs) - 1} arrangements)")
