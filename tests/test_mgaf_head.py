"""MGAF-3DSSD dense head (BASELINE configs[3]) — the harness's target assignment and losses (fv2p_harness/mgaf_model.py) against
fixtures written by the reference's own CenterTargetAssigner and CenterAFHeadTemplate (oracle/gen_golden_center_head.py):

  pyref_center_targets.npz   heat map, indices, masks, regression targets from the reference class; exact for the integer outputs
                             and the float32 heat map, 1e-6 for the float targets (the segmentation map is cv2 in the reference
                             and not in the fixture: it is held by its own properties below)
  pyref_center_losses.npz    the eight loss terms and the gradients of the seven head maps from the reference's get_loss

and the whole MGAFDetector step on the HIP ops against the same Python on the host (oracle backend, DCN forward / backward
answered by oracle/dcn_oracle.py) at a reduced map size."""
import os

import numpy as np
import pytest
import torch

from fv2p_harness import mgaf_model as mm
from oracle.backend import oracle_backend

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cfg_of(fix):
    return type("Cfg", (mm.MGAFConfig,), {"point_cloud_range": tuple(float(v) for v in fix["point_cloud_range"]),
                                          "voxel_size": tuple(float(v) for v in fix["voxel_size"])})


def check_targets(dev):
    fix = np.load(os.path.join(GOLD, "pyref_center_targets.npz"))
    tg = mm.center_targets(torch.from_numpy(fix["gt_boxes"]).to(dev), cfg_of(fix), 3)
    assert tuple(tg["hm_target"].shape) == fix["hm_target"].shape
    assert np.array_equal(tg["ind_target"].cpu().numpy(), fix["ind_target"])
    assert np.array_equal(tg["mask_target"].cpu().numpy(), fix["mask_target"])
    assert np.array_equal(tg["xsys_target"].cpu().numpy(), fix["xsys_target"])
    assert np.array_equal(tg["src_box_target"].cpu().numpy(), fix["src_box_target"])
    # exp() of the two float64 libraries may differ in the last bit before the rounding to float32
    assert np.abs(tg["hm_target"].cpu().numpy() - fix["hm_target"]).max() <= 1e-7
    assert np.array_equal(tg["hm_target"].cpu().numpy() == 1, fix["hm_target"] == 1)          # the peaks the focal loss counts
    assert np.abs(tg["anno_box_target"].cpu().numpy() - fix["anno_box_target"]).max() <= 1e-6
    assert int(fix["mask_target"].sum()) == 22
    return tg, fix


def test_center_targets_equal_the_reference_assigner():
    tg, fix = check_targets("cpu")
    # segmentation map (unpinned: cv2 in the reference): every object's rounded centre pixel is foreground, the map is binary,
    # and its area is within the footprint areas' sum (edge pixels add at most a one-pixel rim)
    segm = tg["segm_target"][:, 0].numpy()
    assert set(np.unique(segm)) <= {0.0, 1.0}
    gt, mask = fix["gt_boxes"], fix["mask_target"]
    for s in range(gt.shape[0]):
        for k in range(mask.shape[1]):
            if mask[s, k]:
                x, y = fix["xsys_target"][s, k].astype(int)
                assert segm[s, y, x] == 1.0
        area = sum(float(gt[s, k, 3] * gt[s, k, 4]) / 0.16 for k in range(min(gt.shape[1], mask.shape[1])) if mask[s, k])   # 0.4 m pixels
        rim = sum(2 * float(gt[s, k, 3] + gt[s, k, 4]) / 0.4 + 4 for k in range(min(gt.shape[1], mask.shape[1])) if mask[s, k])
        assert 0 < segm[s].sum() <= area + rim


def check_losses(dev, tol_loss, tol_grad):
    fix = np.load(os.path.join(GOLD, "pyref_center_losses.npz"))
    cfg = cfg_of(fix)
    names = [n for n, _ in mm.MGAFConfig.heads]
    preds = {n: torch.from_numpy(fix["pred_" + n]).to(dev).requires_grad_(True) for n in names}
    tg = mm.center_targets(torch.from_numpy(fix["gt_boxes"]).to(dev), cfg, 3)
    tg["segm_target"] = torch.from_numpy(fix["segm_target"]).to(dev)
    loss, terms = mm.center_losses(preds, tg, cfg)
    for n, v in terms.items():
        if n.startswith("_"):      # bookkeeping of the term (the scored heat-map cells), not a loss
            continue
        want = float(fix["term_" + n])
        assert abs(float(v.detach()) - want) <= tol_loss * max(1.0, abs(want)), (n, float(v.detach()), want)
    assert abs(float(loss.detach()) - float(fix["loss"])) <= tol_loss * abs(float(fix["loss"]))
    grads = torch.autograd.grad(loss, [preds[n] for n in names])
    for n, g in zip(names, grads):
        want = fix["grad_" + n]
        err = np.linalg.norm(g.cpu().numpy().astype(np.float64) - want) / max(np.linalg.norm(want), 1e-12)
        assert err <= tol_grad, (n, err)


def test_center_losses_equal_the_reference_head():
    with oracle_backend():
        check_losses("cpu", 1e-5, 1e-5)


@pytest.mark.gpu
def test_center_targets_and_losses_on_the_gpu(gpu):
    check_targets(gpu)
    check_losses(gpu, 1e-5, 1e-4)


class SmallMGAF(mm.MGAFConfig):
    """Reduced step: quarter range (BEV map 48 x 44), the yaml's layer list with two convs per level instead of five."""
    point_cloud_range = (0.0, -9.6, -3.0, 17.6, 9.6, 1.0)
    grid_size = (352, 384, 40)
    layer_nums = (2, 2, 2)


def small_inputs(cloud_seed=90):
    import oracle
    from fv2p_harness import synth
    from fv2p_harness.backbone import mean_vfe
    rng = np.array(SmallMGAF.point_cloud_range, np.float32)
    feats, coords, boxes = [], [], []
    for b in range(2):
        pts, bx = synth.lidar_cloud(cloud_seed + b, 3000, pc_range=rng, return_boxes=True)
        v, c, k = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, rng, 5, 16000)
        feats.append(mean_vfe(torch.from_numpy(v), torch.from_numpy(k)))
        coords.append(torch.from_numpy(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1)))
        boxes.append(bx)
    g = max(len(b) for b in boxes)
    gt = np.zeros((2, g, 8), np.float32)
    for i, bx in enumerate(boxes):
        gt[i, :len(bx), :7] = bx
        gt[i, :len(bx), 7] = 1 + (np.arange(len(bx)) % 3)
    return torch.cat(feats), torch.cat(coords), torch.from_numpy(gt)


def mgaf_group(name):
    """Coarse module groups (a group's statistic is the worst / the median parameter in it; finer groups are printed, not judged)."""
    if name.startswith("dense_head."):
        return ".".join(name.split(".")[:2])
    return name.split(".")[0]


def zero_gradient(name):
    """A conv bias in front of train-mode BatchNorm: analytically zero gradient, every run holds rounding noise only."""
    return name.startswith("backbone_3d.") and name.endswith((".conv1.bias", ".conv2.bias"))


def host_runs(seed=0, cloud_seed=90):
    """The step on the host twice: float32 (the oracle run every integer output is compared with) and float64 (the calibration run of
    tests/f64_calibration.py: same modules, same oracle call table, float paths in double)."""
    from oracle.spconv_cpu import cpu_mirror
    torch.manual_seed(seed)
    model = mm.MGAFDetector(SmallMGAF)
    feats, coords, gt = small_inputs(cloud_seed)
    ref, ref64 = cpu_mirror(model), cpu_mirror(model).double()
    with oracle_backend():
        for net, cast in ((ref, lambda t: t), (ref64, lambda t: t.double())):
            net.taps = {}
            # the one discrete choice inside the loss - which 24 heat-map peaks per sample the IoU-score term scores - is the float32
            # oracle run's in all three runs (two peaks within rounding of each other swap freely; each run's OWN choice is compared
            # with it separately): the gradients compared are then gradients of the same function
            net.iou_peaks = None if net is ref else ref.taps["terms"]["_iou_peaks"]
            net(cast(feats), coords, 2, cast(gt)).backward()
    return model, ref, ref64, (feats, coords, gt)


def trainable_grads(net):
    # the DCN layers' frozen bias is added, never trained (modules/modulated_deform_conv.py:38-41)
    return {k: p.grad for k, p in net.named_parameters() if p.requires_grad and p.grad is not None}


def test_float64_host_run_is_the_same_step_and_float32_is_a_percent_away_from_it():
    """What the calibrated bounds rest on, checked without a GPU: the float64 host run takes the float32 run's decisions (targets bit
    for bit, same top-24 peak cells up to near-ties), and the float32 run's gradients sit 1e-3 ... 5e-2 from it upstream of the heads -
    the amplification that makes a hand-set bound a coin."""
    import f64_calibration as cal
    _, ref, ref64, _ = host_runs()
    for k in ("ind_target", "mask_target", "segm_target"):
        assert torch.equal(ref.taps["targets"][k], ref64.taps["targets"][k].to(ref.taps["targets"][k].dtype)), k
    assert next(ref64.parameters()).dtype == torch.float64 and ref64.taps["preds"]["hm"].dtype == torch.float64
    d = cal.distances(trainable_grads(ref), trainable_grads(ref64), zero_gradient)
    up = [v for n, v in d.items() if n.startswith(("backbone_3d.", "backbone_2d."))]
    assert len(up) > 100 and 1e-3 < max(up) < 5e-2, (len(up), max(up))
    assert max(v for n, v in d.items() if n.startswith("dense_head.heads.")) < 1e-2


@pytest.mark.gpu
def test_mgaf_step_matches_cpu_oracle(gpu):
    """MGAFDetector (VoxelResBackBone8x, DCNBEVBackbone, CenterAFHead with the DCNv2 feature adaption, target assignment, eight
    loss terms) forward + backward on the HIP ops against the same modules on the host.  Integer outputs (target maps, indices,
    masks) bit-exact against the float32 oracle run.  Floats - head maps, loss terms, every parameter gradient - by the float64-calibrated
    criterion of tests/f64_calibration.py: the HIP run may be at most K times as far from the host float64 run as the host float32
    oracle run is (median parameter and pooled vector per module group; floor 1e-4; every parameter within 0.25).  No hand-set bound."""
    import f64_calibration as cal
    model, ref, ref64, (feats, coords, gt) = host_runs()
    net = model.to(gpu)
    net.taps = {}
    peaks = ref.taps["terms"]["_iou_peaks"]
    net.iou_peaks = peaks.to(gpu)     # the IoU-score term's cells: the float32 oracle run's in all three runs (host_runs)
    from conftest import deterministic_libraries
    with deterministic_libraries():   # run-to-run identical on one box (MIOpen / rocBLAS atomics off); box-to-box it is not, hence f64
        loss_g = net(feats.to(gpu), coords.to(gpu), 2, gt.to(gpu))
        loss_g.backward()
    assert int(ref.taps["targets"]["mask_target"].sum()) > 0
    for k in ("ind_target", "mask_target", "segm_target", "hm_target"):
        assert torch.equal(net.taps["targets"][k].cpu(), ref.taps["targets"][k]), k
    # the HIP run's OWN choice of cells (top 24 heat-map peaks per sample): the oracle run's up to swaps of near-equal peaks
    with torch.no_grad():
        own = mm.center_losses({k: v.detach() for k, v in net.taps["preds"].items()}, net.taps["targets"], SmallMGAF)[1]["_iou_peaks"].cpu()
    same = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(peaks, own))
    assert same >= peaks.numel() - 4, (same, peaks.numel())
    rel_max = lambda a, t: float((a.detach().cpu().double() - t.detach().double()).abs().max() / t.detach().double().abs().max())
    for name, want in ref64.taps["preds"].items():
        d_hip, d_ref = rel_max(net.taps["preds"][name], want), rel_max(ref.taps["preds"][name], want)
        print(f"head map {name:9s} max-rel distance to float64: hip {d_hip:.2e}  host32 {d_ref:.2e}")
        assert d_hip <= max(cal.K * d_ref, cal.FLOOR), (name, d_hip, d_ref)
    for name, want in ref64.taps["terms"].items():
        if name.startswith("_"):
            continue
        got, want, host = float(net.taps["terms"][name].detach()), float(want.detach()), float(ref.taps["terms"][name].detach())
        assert abs(got - want) <= max(cal.K * abs(host - want), cal.FLOOR * max(1.0, abs(want))), (name, got, host, want)
    rows, bad = cal.compare(trainable_grads(net), trainable_grads(ref), trainable_grads(ref64), mgaf_group, zero_gradient)
    print(cal.report(rows))
    fine = lambda n: ".".join(n.split(".")[:3 if n.startswith(("backbone_2d", "dense_head")) else 2])
    print(cal.report(cal.compare(trainable_grads(net), trainable_grads(ref), trainable_grads(ref64), fine, zero_gradient)[0],
                     "the same by fine group (printed, not judged)"))
    assert not bad, "\n".join(bad)
