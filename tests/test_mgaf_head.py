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


def small_inputs():
    import oracle
    from fv2p_harness import synth
    from fv2p_harness.backbone import mean_vfe
    rng = np.array(SmallMGAF.point_cloud_range, np.float32)
    feats, coords, boxes = [], [], []
    for b in range(2):
        pts, bx = synth.lidar_cloud(90 + b, 3000, pc_range=rng, return_boxes=True)
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


@pytest.mark.gpu
def test_mgaf_step_matches_cpu_oracle(gpu):
    """MGAFDetector (VoxelResBackBone8x, DCNBEVBackbone, CenterAFHead with the DCNv2 feature adaption, target assignment, eight
    loss terms) forward + backward on the HIP ops against the same modules on the host: target maps bit-exact, head maps 1e-3
    (through the 21-layer sparse backbone, the DCN BEV backbone and their BatchNorms), all eight loss terms 1e-3 (the IoU-score term on
    the host run's peak cells when the two runs' top-24 sets differ), every parameter gradient by relative L2 with the per-group bounds
    stated where they are applied: 3e-3 heads, 6e-3 feature adaption, 2e-2 upstream of it, 3e-2 third BEV level."""
    from oracle.spconv_cpu import cpu_mirror
    torch.manual_seed(0)
    model = mm.MGAFDetector(SmallMGAF)
    ref = cpu_mirror(model)
    feats, coords, gt = small_inputs()
    ref.taps = {}
    with oracle_backend():
        loss_c = ref(feats, coords, 2, gt)
        loss_c.backward()
    net = model.to(gpu)
    net.taps = {}
    from conftest import deterministic_libraries
    with deterministic_libraries():   # (without: heads 1.7e-3 ... 3.3e-3, upstream 1.3e-2 ... 2.1e-2 over six runs of this test)
        loss_g = net(feats.to(gpu), coords.to(gpu), 2, gt.to(gpu))
        loss_g.backward()
    assert int(ref.taps["targets"]["mask_target"].sum()) > 0
    for k in ("ind_target", "mask_target", "segm_target", "hm_target"):
        assert torch.equal(net.taps["targets"][k].cpu(), ref.taps["targets"][k]), k
    for name, want in ref.taps["preds"].items():
        got = net.taps["preds"][name].detach().cpu()
        assert float((got - want.detach()).abs().max()) <= 1e-3 * float(want.detach().abs().max()), name
    peaks_c, peaks_g = ref.taps["terms"]["_iou_peaks"], net.taps["terms"]["_iou_peaks"].cpu()
    for name, want in ref.taps["terms"].items():
        if name.startswith("_"):
            continue
        got, want = float(net.taps["terms"][name].detach()), float(want.detach())
        if name == "iouscore" and not torch.equal(peaks_c, peaks_g):
            # The IoU-score term scores the 24 highest heat-map peaks per sample; two peaks within float32 noise of each other may swap
            # between the runs.  Tie-robust form: the HIP run's term re-evaluated on the HOST run's cells (same 1e-3 as every other
            # term), and the two runs must agree on all but a few cells.
            same = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(peaks_c, peaks_g))
            assert same >= peaks_c.numel() - 4, (same, peaks_c.numel())
            with torch.no_grad():
                got = float(mm.center_losses({k: v.detach() for k, v in net.taps["preds"].items()}, net.taps["targets"], SmallMGAF,
                                             peaks=peaks_c.to(gpu))[1]["iouscore"])
        assert abs(got - want) <= 1e-3 * max(1.0, abs(want)), (name, got, want)
    gp = dict(net.named_parameters())
    worst, bad, by_group = ("", 0.0), [], {}
    for name, p in ref.named_parameters():
        if not p.requires_grad:          # the DCN layers' frozen bias (modules/modulated_deform_conv.py:38-41: added, never trained)
            continue
        assert p.grad is not None and gp[name].grad is not None, name
        if name.startswith("backbone_3d.") and name.endswith((".conv1.bias", ".conv2.bias")):
            continue                     # a conv bias in front of train-mode BatchNorm: analytically zero gradient
        a, b = gp[name].grad.cpu().double(), p.grad.double()
        err = float((a - b).norm() / b.norm().clamp_min(1e-12))
        worst = max(worst, (name, err), key=lambda t: t[1])
        grp = ".".join(name.split(".")[:3 if name.startswith(("backbone_2d", "dense_head")) else 2])
        by_group[grp] = max(by_group.get(grp, 0.0), err)
        # Measured per module group (HIP against the host run, printed below; the same numbers in every run under the deterministic
        # library settings): the seven heads <= 2.2e-3, the head's deformable feature adaption 3.9e-3, its offset / mask predictor 1.2e-2
        # (gradients through the bilinear kernel's kinks, where float32 and the oracle's float64 pick sides), shared conv + first two BEV
        # levels + sparse backbone 0.9 ... 1.7e-2 (everything that has crossed the DCN backward and a chain of train-mode BatchNorms: two
        # float32 implementations, tests/test_fv2p_step_gpu.py DEEP_END), third BEV level 1.9e-2 - plain torch convolutions on BOTH
        # sides there, MIOpen on the GPU and oneDNN on the host.
        # Round 3 allowed 6e-2 / 3e-3 / 3e-2 (float atomics in the DCN data gradient moved the heads by 1 ... 2e-3 from run to run;
        # the backward is bit-reproducible now).
        if name.startswith("dense_head.heads."):
            tol = 3e-3
        elif name.startswith("dense_head.feature_adapt.conv_adaption"):
            tol = 6e-3
        elif name.startswith(("backbone_2d.blocks.2", "backbone_2d.deblocks.2")):
            tol = 3e-2
        else:
            tol = 2e-2
        if err >= tol:
            bad.append((name, f"{err:.2e}"))
    print("worst MGAF gradient:", worst)
    print("largest gradient error per module group:", {k: f"{v:.1e}" for k, v in by_group.items()})
    assert not bad, " ".join(f"{n}={e}" for n, e in bad)
