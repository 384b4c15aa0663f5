#!/usr/bin/env python3
"""The roofline conv (subm 128 -> 128, res4 of VoxelResBackBone8x at batch 3) with its rows RENUMBERED so that the XCD-major tiles are
y bands: FV2P_ORDER=ybzx | bzyx (as the rulebook emits them) | shuffle.  Runs the one launch 100 times (for rocprofv3 --pmc FETCH_SIZE:
bash tools/pmc_generic.sh out.json "conv_rows_ksplit<128" "FETCH_SIZE" -- python3 tools/order_traffic.py) and prints its time."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "from-voxel-to-point_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fv2p_harness import synth  # noqa: E402
from fv2p_harness.backbone import VoxelResBackBone8x, mean_vfe  # noqa: E402
from pcdet.datasets.processor.voxel_generator import points_to_voxel_gpu  # noqa: E402
from pcdet.ops.spconv import ops  # noqa: E402
from pcdet.ops.spconv.conv import SparseConvolution  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = VoxelResBackBone8x(4, [1408, 1600, 40]).to(dev)
feats, coords = [], []
for b in range(3):
    v, c, n = points_to_voxel_gpu(torch.from_numpy(synth.lidar_cloud(b, 16384)).to(dev), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 16000)
    feats.append(mean_vfe(v, n))
    coords.append(torch.nn.functional.pad(c, (1, 0), value=b))
recs = []


def hook(mod, inp, out):
    x = inp[0]
    if mod.indice_key is not None:
        recs.append((mod, x.features.detach(), x.indice_dict[mod.indice_key], out.features.shape[0]))


hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, SparseConvolution)]
with torch.no_grad():
    model(torch.cat(feats), torch.cat(coords), 3)
for h in hs:
    h.remove()
mod, f, rb, n_out = max(recs, key=lambda r: int(r[2].indice_pair_num.sum().item()) * r[0].in_channels * r[0].out_channels)
order = os.environ.get("FV2P_ORDER", "bzyx")
ind = rb.indices
inp = ind.cpu().numpy().astype(np.int64)   # (b, z, y, x)
perm = {"bzyx": np.arange(inp.shape[0]), "ybzx": np.lexsort((inp[:, 3], inp[:, 1], inp[:, 0], inp[:, 2])),
        "shuffle": np.random.default_rng(0).permutation(inp.shape[0])}[order]
pt = torch.from_numpy(perm).to(dev)
ind_p, f_p = ind[pt].contiguous(), f[pt].contiguous()
rb_p = ops.build_rulebook(ind_p, 4, rb.spatial_shape, mod.kernel_size, mod.stride, mod.padding, mod.dilation, 0, True)
w = mod.weight.detach()
for _ in range(5):
    ops.indice_conv(f_p, w, rb_p, None, n_out, False, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    ops.indice_conv(f_p, w, rb_p, None, n_out, False, True)
e1.record()
torch.cuda.synchronize()
print(f"rows {order}: subm {mod.in_channels}->{mod.out_channels} n={f.shape[0]} pairs={int(rb.indice_pair_num.sum().item())}: {e0.elapsed_time(e1) * 10:.1f} us per launch")
