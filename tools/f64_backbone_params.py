"""Per-parameter distance to a float64 host run, HIP run and host float32 oracle run, for the two sparse backbones at BASELINE configs[1] size.
    python tools/f64_backbone_params.py [VoxelBackBone8x|VoxelResBackBone8x]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from fv2p_harness import backbone  # noqa: E402
from oracle.spconv_cpu import cpu_mirror  # noqa: E402
from test_backbone_gpu import make_batch  # noqa: E402

gpu = torch.device("cuda:0")
cls = getattr(backbone, sys.argv[1] if len(sys.argv) > 1 else "VoxelBackBone8x")
torch.manual_seed(0)
model = cls(4, [1408, 1600, 40]).to(gpu)
ref, ref64 = cpu_mirror(model), cpu_mirror(model).double()
feats, coords = make_batch(gpu, [3, 4, 5, 6], 16384)
out, _ = model(feats, coords, 4)
o32, _ = ref(feats.cpu(), coords.cpu(), 4)
o64, _ = ref64(feats.cpu().double(), coords.cpu(), 4)
g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(1))
(out.features * g.to(gpu)).sum().backward()
(o32.features * g).sum().backward()
(o64.features * g.double()).sum().backward()
d = lambda a, t: float((a.detach().cpu().double() - t.double()).norm() / t.double().norm().clamp_min(1e-300))
gh, g32 = dict(model.named_parameters()), dict(ref.named_parameters())
print(f"{'parameter':32s} {'|grad|':>10s} {'hip':>9s} {'host32':>9s} {'x':>7s}")
for n, p in ref64.named_parameters():
    if p.grad is None:
        continue
    a, b = d(gh[n].grad, p.grad), d(g32[n].grad, p.grad)
    print(f"{n:32s} {float(p.grad.norm()):10.3e} {a:9.2e} {b:9.2e} {a / max(b, 1e-300):7.1f}")
