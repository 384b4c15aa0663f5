"""Child process of tests/test_fv2p_step_gpu.py::test_stream_arrangements_give_the_same_step (not collected by pytest)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

# Library non-determinism, measured with tools/bev_repro.py (profiles/r04_bev_repro_*.txt): between two IDENTICAL forward passes of one
# process MIOpen's training-mode BatchNorm2d (first differing module: backbone_2d.blocks.0.2) or the 1 x 1 convolutions' split-K GEMMs
# (dense_head.conv_cls) differ in the last bits - which of the two depends on the process - and every later layer inherits it.  With
# MIOpen's deterministic solvers and rocBLAS atomics off, 0 of 186 module outputs differ across all arrangements, and what is left in the
# gradients is this repo's own float-atomic interpolation gradient (2e-6 ... 3e-6 of a gradient's norm).  So the check runs under those
# settings and holds every arrangement to bit-identical forward taps and 1e-4 of each gradient's norm: a missing event or wait does not
# hide behind a library's reassociation any more.
torch.backends.cudnn.deterministic = True
torch.use_deterministic_algorithms(True, warn_only=True)

from fv2p_harness.fv2p_model import FV2PDetector  # noqa: E402
from test_fv2p_step_gpu import SmallFV2P, make_inputs  # noqa: E402

gpu = torch.device("cuda:0")
torch.manual_seed(3)
model = FV2PDetector(SmallFV2P).to(gpu)
clouds, feats, coords, gt, u = make_inputs(SmallFV2P, 2, 4096)
args = ([c.to(gpu) for c in clouds], feats.to(gpu), coords.to(gpu), gt.to(gpu), u.to(gpu))
# the detector's first SAFE_FIRST_STEPS GPU steps run on the calling stream only whatever arrangement is asked for (fv2p_model.forward):
# MIOpen's first-call solver search on a side stream is the one thing that was ever seen to hang the dense-branch arrangement
from fv2p_harness import fv2p_model  # noqa: E402
model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": True, "point_branch_stream": True})
for _ in range(fv2p_model.SAFE_FIRST_STEPS):
    model(*args).backward()
torch.cuda.synchronize()
assert model._gpu_steps == fv2p_model.SAFE_FIRST_STEPS
runs = []
compared = 0
for dense, point in ((True, True), (False, True), (False, False)):
    model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": dense, "point_branch_stream": point})
    model.taps = {}
    model.zero_grad(set_to_none=True)
    torch.manual_seed(11)   # the RoI head's dropout masks: the same in every arrangement
    loss = model(*args)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((loss.item(), model.taps["keypoints"].clone(), model.taps["sampled_rois"].clone(),
                 {k: p.grad.clone() for k, p in model.named_parameters()},
                 (float(model.taps["loss_rpn"]), float(model.taps["loss_point"])), model.taps["prop_scores"].clone()))
names = ("dense branch on a side stream (bench.py's arrangement)", "point branch on a side stream", "one stream")
GRAD_TOL = 1e-4   # of the gradient's norm (the interpolation gradient's float atomics: 3e-6 measured)
loose = 0
for name, other in zip(names[1:], runs[1:]):
    assert torch.equal(other[1], runs[0][1]), f"{name}: other key points than with the {names[0]}"
    identical = torch.equal(other[2], runs[0][2]) and torch.equal(other[5], runs[0][5]) and other[4] == runs[0][4] and other[0] == runs[0][0]
    if not identical:
        # not seen under the deterministic library settings above; if a box ever shows it, say so loudly and fall back to the loose bounds
        # (the forward pass then carries a library's run-to-run noise and the second stage may sample other RoIs)
        loose += 1
        print(f"WARNING {name}: forward pass not bit-identical to the {names[0]} (losses {other[0]!r} / {runs[0][0]!r}, "
              f"proposal scores differ by {float((other[5] - runs[0][5]).abs().max()):.2e}): loose bounds for this arrangement")
        for what, a, b in zip(("anchor-head loss", "point-head loss"), other[4], runs[0][4]):
            assert abs(a - b) < 1e-5 * max(1.0, abs(b)), f"{name}: {what} {a} against {b}"
        if other[2].shape != runs[0][2].shape or float((other[2] - runs[0][2]).abs().max()) > 1e-3:
            continue
    compared += 1
    assert abs(other[0] - runs[0][0]) <= (0.0 if identical else 1e-4 * max(1.0, abs(runs[0][0]))), f"{name}: loss {other[0]} against {runs[0][0]}"
    for k, g0 in runs[0][3].items():
        rel = GRAD_TOL if identical else (2e-2 if k.startswith(("backbone_3d.", "post_pfe.")) else 2e-3)
        # (the bias of a conv that feeds BatchNorm has a zero gradient up to rounding: absolute floor beside the relative bound)
        err, bound = float((other[3][k] - g0).norm()), rel * float(g0.norm()) + 1e-6 * g0.numel() ** 0.5
        assert err < bound, f"{name}: gradient of {k} differs by {err:.3e} (bound {bound:.3e})"
assert compared == len(runs) - 1 or loose, "an arrangement was skipped although its forward pass was bit-identical"
assert compared >= 1, "no arrangement sampled the same RoIs as the first: nothing of the second stage was compared"
assert loose == 0 or os.environ.get("FV2P_ALLOW_LOOSE_ARRANGEMENTS") == "1", \
    f"{loose} arrangement(s) had a forward pass that was not bit-identical under deterministic library settings: investigate (tools/bev_repro.py)"
print(f"ARRANGEMENTS AGREE (second stage compared in {compared} of {len(runs) - 1} arrangements, forward passes bit-identical: {loose == 0})")
