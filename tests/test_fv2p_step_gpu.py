"""FV2P training-step replay (BASELINE configs[2]) at reduced size: the harness detector on the HIP ops versus the very same
Python run on the CPU with every C-ABI call answered by the oracle (oracle/backend.py) and every sparse conv by the
oracle's gather-mm-scatter (oracle/spconv_cpu.py).

Integer outputs are compared bit for bit (key points chosen by FPS, NMS survivors, sampled RoIs); float features within
1e-3 relative through the 21-layer residual backbone + decoder (per-op tolerance 1e-4, see test_backbone_gpu.py), losses
within 1e-3.  The second stage is additionally checked in isolation on identical inputs, because its proposals are a
top-k + NMS over network outputs, where a 1e-6 score difference between two runs may legitimately swap neighbours."""
import numpy as np
import pytest
import torch

import oracle
from fv2p_harness import synth
from fv2p_harness.backbone import mean_vfe
from fv2p_harness.fv2p_model import FV2PConfig, FV2PDetector, FV2PWaymoConfig, pad_gt_boxes
from oracle.backend import oracle_backend
from oracle.spconv_cpu import cpu_mirror


class SmallFV2P(FV2PConfig):
    """Half the KITTI range (BEV map 100 x 88), 2048 key points, 1024 -> 128 proposals, 32 RoIs x 128 pooled points."""
    point_cloud_range = (0.0, -20.0, -3.0, 35.2, 20.0, 1.0)
    grid_size = (704, 800, 40)
    num_keypoints = 2048
    nms_pre, nms_post = 1024, 128
    roi_per_image = 32
    num_sampled_points = 128
    dp_ratio = 0.0          # dropout draws differ between devices


GRAD_TOL = 2e-3     # relative L2 per parameter gradient where both sides see IDENTICAL inputs and a short chain (the RoI head test, the
#                     reference-call-structure test: two runs on the same GPU)
# The first-stage gradients of the whole step are held by the float64-calibrated criterion of tests/f64_calibration.py (round 5): the
# host run is repeated in float64 (the one float32-only layer on that path, the BEV gather, follows the dtype of a HOST tensor; CUDA
# tensors are float32 or refused as before) and the HIP run may be at most K times as far from it as the float32 oracle run is, per module
# group.  Rounds 3 - 4 had hand-set bounds here (2e-2, then 6e-3 for the sparse backbone + decoder against a measured 4.7e-3); the
# measured number turned out to depend on the HOST: 4.67e-3 with 128 or 32 BLAS threads, 5.86e-3 with 6, 1.56e-2 with one
# (profiles/r05_chain_tests_host_threads.txt) - the float32 oracle run moves, not the HIP run.
DEEP_END = ("backbone_3d.", "post_pfe.")
DEEP_END_TOL = 1e-2     # only for test_reference_call_structure_is_the_same_step (HIP against HIP: no host in the comparison)


def grad_tol(name):
    return DEEP_END_TOL if name.startswith(DEEP_END) else GRAD_TOL


def zero_gradient(name):
    """A conv bias in front of train-mode BatchNorm: its gradient is analytically zero, both sides hold rounding noise."""
    return name.startswith("backbone_3d.") and name.endswith((".conv1.bias", ".conv2.bias"))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def make_inputs(cfg, batch, n_points):
    rng = np.array(cfg.point_cloud_range, np.float32)
    clouds, boxes, feats, coords = [], [], [], []
    for b in range(batch):
        pts, bx = synth.lidar_cloud(40 + b, n_points, pc_range=rng, return_boxes=True)
        clouds.append(torch.from_numpy(pts))
        boxes.append(bx)
        v, c, k = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, rng, 5, 16000)
        feats.append(mean_vfe(torch.from_numpy(v), torch.from_numpy(k)))
        coords.append(torch.from_numpy(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1)))
    gt = pad_gt_boxes(boxes, "cpu")
    u = torch.rand(batch, cfg.nms_post + cfg.roi_per_image, generator=torch.Generator().manual_seed(1))
    return clouds, torch.cat(feats), torch.cat(coords), gt, u


@pytest.fixture(scope="module")
def cpu_run():
    torch.manual_seed(0)
    model = FV2PDetector(SmallFV2P)
    ref = cpu_mirror(model)
    ref.taps = {}
    inputs = make_inputs(SmallFV2P, 2, 4096)
    with oracle_backend():
        loss = ref(*inputs)
        # first-stage + point losses first (their gradients are compared with the GPU run), then the second stage on top
        (ref.taps["loss_rpn"] + ref.taps["loss_point"]).backward(retain_graph=True)
        stage1 = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
        ref.taps["loss_rcnn"].backward()
        # the same first stage in float64: the calibration run (tests/f64_calibration.py)
        ref64 = cpu_mirror(model).double()
        ref64.taps = {}
        clouds, feats, coords, gt, u = inputs
        ref64(clouds, feats.double(), coords, gt, u)
        (ref64.taps["loss_rpn"] + ref64.taps["loss_point"]).backward()
        ref.stage1_f64 = {k: p.grad.clone() for k, p in ref64.named_parameters() if p.grad is not None}
        ref.taps_f64 = ref64.taps
    return model, ref, inputs, loss, stage1


def test_cpu_replay_runs_and_trains_every_parameter(cpu_run):
    _, ref, _, loss, _ = cpu_run
    assert torch.isfinite(loss)
    missing = [k for k, p in ref.named_parameters() if p.grad is None]
    assert not missing, missing
    t = ref.taps
    assert t["keypoints"].shape == (2, 2048, 3) and t["rois"].shape == (2, 128, 7) and t["sampled_rois"].shape == (2, 32, 7)
    assert (t["roi_iou"] >= 0).all() and (t["roi_iou"] <= 1).all()


@pytest.mark.gpu
def test_fv2p_step_matches_cpu_oracle(gpu, cpu_run):
    model, ref, inputs, ref_loss, stage1 = cpu_run
    net = model.to(gpu)
    net.taps = {}
    clouds, feats, coords, gt, u = inputs
    from conftest import deterministic_libraries
    with deterministic_libraries():   # the GPU side is then the same number in every run (conftest.py)
        loss = net([c.to(gpu) for c in clouds], feats.to(gpu), coords.to(gpu), gt.to(gpu), u.to(gpu))
        g, c = net.taps, ref.taps
        # gradients of the first-stage + point losses only: the second stage's RoIs are a top-k + NMS over network outputs and may
        # legitimately differ between two float implementations (it is compared on identical inputs in the next test)
        (g["loss_rpn"] + g["loss_point"]).backward(retain_graph=True)
    assert torch.equal(g["keypoints"].cpu(), c["keypoints"])                          # FPS order: bit-exact
    import f64_calibration as cal
    t64 = ref.taps_f64
    for name in ("point_features", "bev"):      # features through the 21-layer backbone (+ decoder / BEV backbone): calibrated like the gradients
        d_hip, d_ref = rel(g[name].detach().cpu(), t64[name].detach()), rel(c[name].detach(), t64[name].detach())
        print(f"{name:15s} max-rel distance to float64: hip {d_hip:.2e}  host32 {d_ref:.2e}")
        assert d_hip <= max(cal.K * d_ref, cal.FLOOR), (name, d_hip, d_ref)
    for name in ("loss_point", "loss_rpn"):
        got, host, want = g[name].item(), c[name].item(), float(t64[name])
        assert abs(got - want) <= max(cal.K * abs(host - want), cal.FLOOR * max(1.0, abs(want))), (name, got, host, want)
    # EVERY parameter the first-stage + point losses reach: the HIP run at most K times as far from the float64 host run as the float32
    # oracle run is, per module group (median and pooled relative L2; floor 1e-4; every parameter within 0.25)
    assert torch.equal(ref.taps_f64["keypoints"].float(), c["keypoints"])          # the float64 run takes the float32 run's decisions
    gp = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    reach = {k: v for k, v in ref.stage1_f64.items() if float(v.norm()) >= 1e-10}  # second-stage layers: not reached by the two losses
    for name in set(ref.stage1_f64) - set(reach):
        if not zero_gradient(name):
            assert float(gp[name].norm()) < 1e-8, name
    group = lambda n: n.split(".")[0]
    # floor: FLIPS - the same 21-layer backbone as tests/test_backbone_gpu.py sits at the deep end of this chain, and a single ReLU decision
    # that differs from the float64 run leaves 1e-4 ... 1e-3 in everything upstream of it (HIP: the decoder's median 7.9e-5 against the host
    # run's 1.3e-6 on the boxes of round 5, a ratio without meaning)
    rows, bad = cal.compare(gp, stage1, reach, group, zero_gradient, floor=cal.FLIPS)
    print(cal.report(rows))
    print(cal.report(cal.compare(gp, stage1, reach, lambda n: ".".join(n.split(".")[:2]), zero_gradient)[0], "the same by sub-module (printed, not judged)"))
    worst = max(((float((gp[k].cpu().double() - v.double()).norm() / v.double().norm()), k) for k, v in stage1.items()
                 if k in reach and not zero_gradient(k)))
    print(f"worst first-stage gradient against the host float32 run (informative): {worst[1]} {worst[0]:.2e}")
    assert not bad, "\n".join(bad)
    g["loss_rcnn"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    # the second-stage loss is compared unconditionally in test_roi_head_on_identical_inputs_matches_cpu_oracle (both heads are fed
    # the host run's proposals there); here the two runs' own proposals may differ by a swapped pair of near-equal scores
    if torch.equal(g["sampled_rois"].cpu(), c["sampled_rois"]):
        assert abs(g["loss_rcnn"].item() - c["loss_rcnn"].item()) < 2e-3 * max(1.0, abs(c["loss_rcnn"].item()))
    # bench.py's cpu_baseline leg mirrors a model that has already stepped on the GPU: its side streams must not be module state
    net.taps = None
    assert next(cpu_mirror(net).parameters()).device.type == "cpu"


@pytest.mark.gpu
def test_roi_head_on_identical_inputs_matches_cpu_oracle(gpu, cpu_run):
    model, ref, inputs, _, _ = cpu_run
    head_g = model.roi_head.to(gpu)
    t = ref.taps
    gt, u = inputs[3], inputs[4]

    def run(head, dev):
        f = t["point_features"].detach().to(dev).clone().requires_grad_(True)
        bev = t["bev"].detach().to(dev).clone().requires_grad_(True)
        loss, aux = head(t["keypoints"].to(dev), f, t["point_scores"].detach().to(dev), bev, t["prop_scores"].to(dev),
                         t["prop_boxes"].to(dev), gt.to(dev), u.to(dev))
        head.zero_grad(set_to_none=True)
        loss.backward()
        return loss, aux, f.grad, bev.grad

    with oracle_backend():
        lc, ac, fc, bc = run(ref.roi_head, "cpu")
    lg, ag, fg, bg = run(head_g, gpu)
    assert torch.equal(ag["rois"].cpu(), ac["rois"])                                  # NMS survivors: bit-exact
    assert torch.equal(ag["sampled_rois"].cpu(), ac["sampled_rois"])
    assert rel(ag["roi_iou"].cpu(), ac["roi_iou"]) < 1e-5
    assert abs(lg.item() - lc.item()) < 1e-3 * max(1.0, abs(lc.item()))
    assert fg is None and fc is None          # the RoI point pool carries no gradient (reference: under no_grad, iouguided_roi_head.py:178)
    assert rel(bg.cpu(), bc) < 2e-3
    gp, cp = dict(head_g.named_parameters()), dict(ref.roi_head.named_parameters())
    for name, p in cp.items():           # every parameter of the head
        if p.grad is None:
            assert gp[name].grad is None, name
            continue
        a, b = gp[name].grad.cpu().double(), p.grad.double()
        assert float((a - b).norm() / b.norm().clamp_min(1e-12)) < GRAD_TOL, name


@pytest.mark.gpu
def test_reference_call_structure_is_the_same_step(gpu, cpu_run):
    """fv2p_harness.refstyle (bench.py's vs_restated_structure leg: per-offset gather -> mm -> scatter-add sparse convs, separate BatchNorm /
    ReLU modules, grouped set abstraction, per-sample NMS, tensor-op target assignment, plain FPS kernel, one stream) computes the
    step of the default path: same key points and proposals, features / losses within 1e-3, gradients within GRAD_TOL."""
    from fv2p_harness import refstyle
    model, _, inputs, _, _ = cpu_run
    clouds, feats, coords, gt, u = inputs
    args = ([c.to(gpu) for c in clouds], feats.to(gpu), coords.to(gpu), gt.to(gpu), u.to(gpu))
    net = model.to(gpu)
    runs = []
    for ref_mode in (False, True):
        net.taps = {}
        net.zero_grad(set_to_none=True)
        net.cfg = refstyle.inline_config(SmallFV2P) if ref_mode else SmallFV2P
        if ref_mode:
            with refstyle.reference_call_structure():
                loss = net(*args)
                (net.taps["loss_rpn"] + net.taps["loss_point"]).backward()
        else:
            loss = net(*args)
            (net.taps["loss_rpn"] + net.taps["loss_point"]).backward()
        runs.append((dict(net.taps), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, loss.detach()))
    net.cfg, net.taps = SmallFV2P, None
    (ta, ga, la), (tb, gb, lb) = runs
    assert torch.equal(ta["keypoints"], tb["keypoints"])
    assert rel(tb["point_features"].detach().cpu(), ta["point_features"].detach().cpu()) < 1e-3
    assert rel(tb["bev"].detach().cpu(), ta["bev"].detach().cpu()) < 1e-3
    for name in ("loss_rpn", "loss_point"):
        assert abs(ta[name].item() - tb[name].item()) < 1e-3 * max(1.0, abs(ta[name].item())), name
    assert set(ga) == set(gb)
    bad = []
    for name, b in ga.items():
        if float(b.norm()) < 1e-10 or zero_gradient(name):
            continue
        err = float((gb[name].double() - b.double()).norm() / b.double().norm())
        if err >= grad_tol(name):
            bad.append((name, f"{err:.2e}"))
    assert not bad, " ".join(f"{n}={e}" for n, e in bad)


class SmallWaymoFV2P(FV2PWaymoConfig):
    """BASELINE configs[4] shape at reduced size: half the Waymo range (BEV map 94 x 94), 0.1 m voxels, five point features,
    30 000-point clouds (> 24 576: the streaming FPS kernel), 2048 key points."""
    point_cloud_range = (-37.6, -37.6, -2.0, 37.6, 37.6, 4.0)
    grid_size = (752, 752, 40)
    num_keypoints = 2048
    nms_pre, nms_post = 1024, 128
    roi_per_image = 32
    num_sampled_points = 128
    dp_ratio = 0.0


@pytest.mark.gpu
def test_waymo_shaped_step_matches_cpu_oracle(gpu):
    cfg = SmallWaymoFV2P
    torch.manual_seed(1)
    model = FV2PDetector(cfg)
    ref = cpu_mirror(model)
    ref.taps = {}
    rng = np.array(cfg.point_cloud_range, np.float32)
    vs = np.array(cfg.voxel_size, np.float32)
    clouds, boxes, feats, coords = [], [], [], []
    for b in range(2):
        pts, bx = synth.lidar_cloud(70 + b, 30000, pc_range=rng, fov_deg=180.0, az_step_deg=0.13, return_boxes=True)
        pts = np.concatenate([pts, np.random.default_rng(b).uniform(0, 1, (pts.shape[0], 1)).astype(np.float32)], 1)
        clouds.append(torch.from_numpy(pts))
        boxes.append(bx)
        v, c, k = oracle.points_to_voxel(pts, vs, rng, 5, cfg.max_voxels)
        feats.append(mean_vfe(torch.from_numpy(v), torch.from_numpy(k)))
        coords.append(torch.from_numpy(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1)))
    gt = pad_gt_boxes(boxes, "cpu")
    u = torch.rand(2, cfg.nms_post + cfg.roi_per_image, generator=torch.Generator().manual_seed(2))
    feats, coords = torch.cat(feats), torch.cat(coords)
    with oracle_backend():
        ref(clouds, feats, coords, gt, u)
    net = model.to(gpu)
    net.taps = {}
    net([c.to(gpu) for c in clouds], feats.to(gpu), coords.to(gpu), gt.to(gpu), u.to(gpu))
    g, c = net.taps, ref.taps
    assert torch.equal(g["keypoints"].cpu(), c["keypoints"])                          # streaming FPS: bit-exact
    assert rel(g["point_features"].detach().cpu(), c["point_features"].detach()) < 1e-3
    assert rel(g["bev"].detach().cpu(), c["bev"].detach()) < 1e-3
    for name in ("loss_rpn", "loss_point"):
        assert abs(g[name].item() - c[name].item()) < 1e-3 * max(1.0, abs(c[name].item())), name
    (g["loss_rpn"] + g["loss_point"] + g["loss_rcnn"]).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.gpu
def test_roi_target_sampling_kernel_equals_its_tensor_formulation(gpu):
    """fv2p_roi_sample_targets (ProposalTargetLayer.subsample_rois, proposal_target_layer.py:92-217, for a batch in one launch)
    against the same rule written in torch ops: sampled RoIs, their boxes and overlaps bit for bit — mixed sets, no foreground,
    no background at all (picks with replacement), no easy / no hard background."""
    from fv2p_harness.fv2p_model import IoUGuidedRoIHead
    import fv2p_native
    head = IoUGuidedRoIHead(SmallFV2P)
    cfg = SmallFV2P
    r, n, g = 128, cfg.roi_per_image, 12
    rng = np.random.default_rng(5)
    cases = []
    base = rng.uniform(0, 1, size=(4, r, g)).astype(np.float32) ** 3          # mostly low overlaps, a few high
    cases.append(base)
    cases.append(np.minimum(base, 0.5).astype(np.float32))                     # no foreground
    cases.append((0.6 + 0.4 * base).astype(np.float32))                        # no background: all picked with replacement
    cases.append(np.where(base < 0.1, 0.2, base).astype(np.float32))           # no easy background
    cases.append(np.where((base >= 0.1) & (base < 0.55), 0.05, base).astype(np.float32))   # no hard background
    cases.append(np.zeros_like(base))                                          # nothing overlaps
    for iou_np in cases:
        iou = torch.from_numpy(iou_np).to(gpu)
        b = iou.shape[0]
        rois = torch.from_numpy(rng.standard_normal((b, r, 7)).astype(np.float32)).to(gpu)
        gt = torch.from_numpy(rng.standard_normal((b, g, 8)).astype(np.float32)).to(gpu)
        u = torch.from_numpy(rng.uniform(0, 1, size=(b, r + n)).astype(np.float32)).to(gpu)
        u[:, 3] = u[:, 7]                                                      # equal keys: the permutation is a stable sort
        want = head.sample_targets_tensor_ops(iou, rois, gt, u)
        s_rois, s_gt = rois.new_empty(b, n, 7), gt.new_empty(b, n, 8)
        s_iou, s_index = rois.new_empty(b, n), torch.empty((b, n), dtype=torch.int32, device=gpu)
        fv2p_native.call("fv2p_roi_sample_targets", iou, rois, gt, u, b, r, g, n, 8, float(min(cfg.reg_fg, cfg.cls_fg)), float(cfg.cls_bg_lo),
                         float(cfg.reg_fg), int(round(cfg.fg_ratio * n)), float(cfg.hard_bg_ratio), s_rois, s_gt, s_iou, s_index, fv2p_native.stream())
        assert torch.equal(s_rois, want[0]) and torch.equal(s_iou, want[2])
        # the box of a RoI whose best overlap is shared by several boxes is the first of them in both formulations
        assert torch.equal(s_gt, want[1])


@pytest.mark.gpu
def test_stream_arrangements_give_the_same_step(gpu):
    """The schedules of the forward pass — dense branch on a side stream (bench.py's choice), point branch on a side stream after the
    RoI preparation (the default), everything on one stream — are the same computation.  The child runs under deterministic library settings (MIOpen deterministic
    solvers, rocBLAS atomics off: the two library sources of run-to-run noise that tools/bev_repro.py isolated), where the forward
    pass is bit-identical across arrangements: equal key points, proposals, sampled RoIs and losses BIT FOR BIT, every gradient within
    1e-4 of its norm (what remains is the interpolation gradient's float atomics, 3e-6).  Like bench.py the child takes its first step on
    the calling stream only (MIOpen's first-call solver search on a side stream is what hung the dense-branch arrangement)
    — a hang after that is a failure, not a skip."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arrangement_check.py")
    try:
        out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.fail("the stream arrangements did not finish within 240 s: a hung device queue is a defect of the arrangement")
    assert out.returncode == 0 and "ARRANGEMENTS AGREE" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_anchor_assignment_kernel_equals_its_tensor_formulation(gpu):
    """fv2p_anchor_assign (AxisAlignedTargetAssigner, axis_aligned_target_assigner.py:66-210, for a batch in two launches) against
    AnchorHead.assign_tensor_ops: labels bit for bit (matched / forced / ignored / background), regression targets 1e-6; with
    zero-padded boxes, a sample without boxes and a box no anchor overlaps."""
    from fv2p_harness.fv2p_model import AnchorHead
    head = AnchorHead(SmallFV2P, 128).to(gpu)
    boxes = []
    for s in range(3):
        _, bx = synth.lidar_cloud(11 + s, 2048, pc_range=np.array(SmallFV2P.point_cloud_range, np.float32), return_boxes=True)
        boxes.append(bx)
    boxes[2] = boxes[2][:0]                                        # a sample without ground truth
    boxes[1] = np.concatenate([boxes[1], np.array([[500.0, 500.0, 0.0, 3.9, 1.6, 1.5, 0.3]], np.float32)])   # outside every anchor
    gt = pad_gt_boxes(boxes, gpu, max_gt=48)
    lab, reg = head.assign(gt)
    lab_t, reg_t = head.assign_tensor_ops(gt)
    assert torch.equal(lab, lab_t)
    assert int((lab > 0).sum()) > 0 and int((lab == -1).sum()) > 0 and int((lab[2] != 0).sum()) == 0
    assert float((reg - reg_t).abs().max()) < 1e-6


@pytest.mark.gpu
def test_anchor_loss_kernel_equals_its_tensor_formulation(gpu):
    """fv2p_anchor_loss (focal + smooth-L1 with the sin-difference heading + direction bins, anchor_head_template.py:98-206) against
    AnchorHead.anchor_losses_tensor_ops: the loss within 1e-5, the gradients of the three logit tensors within 1e-4 (relative L2),
    also for a sample without positive anchors."""
    from fv2p_harness.fv2p_model import AnchorHead, AnchorLossFn
    torch.manual_seed(4)
    head = AnchorHead(SmallFV2P, 128).to(gpu)
    boxes = []
    for s in range(3):
        _, bx = synth.lidar_cloud(21 + s, 2048, pc_range=np.array(SmallFV2P.point_cloud_range, np.float32), return_boxes=True)
        boxes.append(bx)
    boxes[1] = boxes[1][:0]
    gt = pad_gt_boxes(boxes, gpu, max_gt=48)
    labels, reg_t = head.assign(gt)
    a = labels.shape[1]
    mk = lambda c, scale: (torch.randn(3, a, c, device=gpu) * scale).requires_grad_(True)
    cls, box, dirs = mk(1, 2.0), mk(7, 0.5), mk(2, 1.0)
    want = head.anchor_losses_tensor_ops(cls, box, dirs, labels, reg_t)
    gw = torch.autograd.grad(want, (cls, box, dirs))
    got = AnchorLossFn.apply(cls, box, dirs, labels, reg_t, head.anchor_rot, SmallFV2P)
    gg = torch.autograd.grad(got * 1.0, (cls, box, dirs))
    assert abs(got.item() - want.item()) < 1e-5 * max(1.0, abs(want.item()))
    for x, y in zip(gg, gw):
        assert float((x - y).norm() / y.norm().clamp_min(1e-12)) < 1e-4


@pytest.mark.gpu
def test_bev_stream_per_grid_column_equals_the_per_point_gather(gpu):
    """IoUGuidedRoIHead.roi_streams: the g grid points of a column share their BEV position, so the harness gathers and compresses once
    per column and broadcasts over z.  Against the reference's form (every one of the g^3 points gathered, bev_grid_pooling.py:68-125;
    fv2p_model.KERNEL_GLUE = False): the compressed BEV stream to 1e-5 (train-mode BatchNorm over repeated rows: the same statistics),
    the gradients of the map and of the compression weights to 1e-4 relative."""
    from fv2p_harness import fv2p_model
    from fv2p_harness.fv2p_model import IoUGuidedRoIHead
    torch.manual_seed(2)
    head = IoUGuidedRoIHead(SmallFV2P).to(gpu)
    b, n = 2, 24
    bev = torch.randn(b, SmallFV2P.bev_pool_in, 100, 88, device=gpu, requires_grad=True)
    rois = torch.cat((torch.rand(b, n, 2, device=gpu) * torch.tensor([30.0, 36.0], device=gpu) + torch.tensor([2.0, -18.0], device=gpu),
                      torch.rand(b, n, 1, device=gpu) - 1.0, torch.rand(b, n, 3, device=gpu) * 2 + 1.5, torch.rand(b, n, 1, device=gpu) * 6.28), dim=-1)
    go = None
    outs = []
    for glue in (False, True):
        fv2p_model.KERNEL_GLUE = glue
        try:
            for p in head.parameters():
                p.grad = None
            bev.grad = None
            g_bev = head.roi_streams(bev, rois)["g_bev"]
            go = torch.randn_like(g_bev) if go is None else go
            g_bev.backward(go)
            outs.append((g_bev.detach().clone(), bev.grad.clone(),
                         {k: p.grad.clone() for k, p in head.bev_grid_pool_layer.named_parameters() if p.grad is not None}))
        finally:
            fv2p_model.KERNEL_GLUE = True
    rel = lambda a, r: float((a - r).abs().max() / r.abs().max().clamp_min(1e-12))
    assert outs[0][0].shape == outs[1][0].shape and rel(outs[1][0], outs[0][0]) < 1e-5
    assert rel(outs[1][1], outs[0][1]) < 1e-4
    assert len(outs[0][2]) >= 2
    for k in outs[0][2]:
        assert rel(outs[1][2][k], outs[0][2][k]) < 1e-4, k
