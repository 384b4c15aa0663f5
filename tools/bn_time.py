"""BatchNorm1d (+ReLU) forward / backward times (events around the Python call: launch overhead included, ~25 us floor): one-launch passes
(grid barrier) / wide reduce finalised by its launch + apply / the <= 64-workgroup reduce + folding apply of rounds 2 - 5."""
import os, sys
import torch, torch.nn as nn
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "from-voxel-to-point_amd")]
import fv2p_native
from pcdet.ops.spconv.norm import batch_norm_relu
e = fv2p_native.torch_ext()
dev = torch.device("cuda")
for n, c in [(35000, 16), (39000, 32), (22000, 64), (10000, 128), (49152, 64), (49152, 128), (8373, 128)]:
    line = f"{n:6d} x {c:3d}:"
    for one, wide in ((True, True), (False, True), (False, False)):
        e.set_bn_one(one)
        e.set_bn_wide(wide)
        bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev)
        x = torch.randn(n, c, device=dev, requires_grad=True)
        g = torch.randn(n, c, device=dev)
        relu = nn.ReLU()
        def fwd():
            return batch_norm_relu(bn, x, relu)
        for _ in range(10):
            y = fwd(); y.backward(g)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        tf = tb = 0.0
        for _ in range(50):
            e0.record(); y = fwd(); e1.record(); y.backward(g); e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
        line += f"  {'one (where it pays)' if one else 'wide reduce + apply' if wide else 'reduce + folding apply'}: fwd {tf / 50 * 1e3:6.1f} us  bwd {tb / 50 * 1e3:6.1f} us |"
    print(line)
e.set_bn_one(True)
e.set_bn_wide(True)
