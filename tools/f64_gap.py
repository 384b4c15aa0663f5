"""How far is the sparse backbone's float32 backward from a float64 run, and which fused piece decides it?  One process = one setting
(the switches are read once per process); tools/f64_gap.sh loops over them.  Prints, per setting, the distance of every parameter
gradient from the float64 run (relative L2) for the HIP path and for torch's float32 formulation of the same network, worst first.
    FV2P_FUSED_BN=0 python tools/f64_gap.py"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from fv2p_harness import refstyle  # noqa: E402
from fv2p_harness.backbone import VoxelResBackBone8x  # noqa: E402
from test_backbone_gpu import make_batch  # noqa: E402

gpu = torch.device("cuda:0")
torch.manual_seed(0)
model = VoxelResBackBone8x(4, [1408, 1600, 40]).to(gpu)
feats, coords = make_batch(gpu, [3, 4], 8192)
g = None


def run(net, x, ref_mode):
    global g
    net.zero_grad(set_to_none=True)
    if ref_mode:
        with refstyle.reference_call_structure():
            out, _ = net(x, coords, 2)
    else:
        out, _ = net(x, coords, 2)
    if g is None:
        g = torch.randn(out.features.shape, device=gpu, generator=torch.Generator(device=gpu).manual_seed(1))
    (out.features * g.to(out.features.dtype)).sum().backward()
    return {k: p.grad.double().clone() for k, p in net.named_parameters() if p.grad is not None}, out.features.detach().double()


native, f_native = run(model, feats, False)
native2, _ = run(model, feats, False)
torch32, f_t32 = run(model, feats, True)
torch64, f_64 = run(copy.deepcopy(model).double(), feats.double(), True)
rows = []
for name, want in torch64.items():
    if name.endswith(("conv1.bias", "conv2.bias")) or float(want.norm()) < 1e-12:
        continue
    rows.append((float((native[name] - want).norm() / want.norm()), float((torch32[name] - want).norm() / want.norm()),
                 float((native2[name] - native[name]).norm() / want.norm()), name))
rows.sort(reverse=True)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("FV2P_")) or "default"
fo = lambda a, b: float((a - b).norm() / b.norm())
print(f"== {tag}: forward output vs float64: HIP {fo(f_native, f_64):.2e}, torch f32 {fo(f_t32, f_64):.2e}")
ratio = sorted(a / max(b, 1e-12) for a, b, _, _ in rows)
print(f"   gradients: HIP/torch ratio of distances to float64: median {ratio[len(ratio) // 2]:.2f}, max {ratio[-1]:.2f}; worst HIP {rows[0][0]:.2e} ({rows[0][3]}), "
      f"worst torch {max(r[1] for r in rows):.2e}; run-to-run of the HIP path (max over parameters) {max(r[2] for r in rows):.1e}")
for a, b, c, n in rows[:6]:
    print(f"     {n:40s} HIP {a:.2e}  torch f32 {b:.2e}")
