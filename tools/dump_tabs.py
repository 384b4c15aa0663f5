import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "from-voxel-to-point_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from fv2p_harness import synth
from fv2p_harness.backbone import VoxelResBackBone8x, mean_vfe
from pcdet.datasets.processor.voxel_generator import points_to_voxel_gpu
from pcdet.ops.spconv.conv import SparseConvolution
dev = torch.device("cuda:0")
model = VoxelResBackBone8x(4, [1408, 1600, 40]).to(dev)
feats, coords = [], []
for b in range(3):
    v, c, n = points_to_voxel_gpu(torch.from_numpy(synth.lidar_cloud(b, 16384)).to(dev), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 16000)
    feats.append(mean_vfe(v, n)); coords.append(torch.nn.functional.pad(c, (1, 0), value=b))
out = {}
def hook(mod, inp, o):
    rb = inp[0].indice_dict[mod.indice_key]
    out[mod.indice_key + "_in"] = rb.tab_in.cpu().numpy().astype(np.int32)
    if rb.tab_out is not None:
        out[mod.indice_key + "_out"] = rb.tab_out.cpu().numpy().astype(np.int32)
hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, SparseConvolution)]
with torch.no_grad():
    model(torch.cat(feats), torch.cat(coords), 3)
keep = {k: v for k, v in out.items() if k.split("_")[0] in ("res3", "res4", "spconv4", "spconv", "res2")}
np.savez_compressed(os.path.join(REPO, "gpurun_out", "tabs_kitti.npz"), **keep)
print({k: v.shape for k, v in keep.items()})
