"""Diagnostic: per-workgroup duration of the planned conv_rows_ksplit launch of the res4 layer against the tile's rows / pairs / MFMA groups."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "from-voxel-to-point_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fv2p_native
from fv2p_harness import synth
from fv2p_harness.backbone import VoxelResBackBone8x, mean_vfe
from pcdet.datasets.processor.voxel_generator import points_to_voxel_gpu
from pcdet.ops.spconv import ops
from pcdet.ops.spconv.conv import SparseConvolution
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = VoxelResBackBone8x(4, [1408, 1600, 40]).to(dev)
feats, coords = [], []
for b in range(3):
    v, c, n = points_to_voxel_gpu(torch.from_numpy(synth.lidar_cloud(b, 16384)).to(dev), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 16000)
    feats.append(mean_vfe(v, n)); coords.append(torch.nn.functional.pad(c, (1, 0), value=b))
recs = []
def hook(mod, inp, out):
    if mod.indice_key is not None:
        recs.append((mod, inp[0].features.detach(), inp[0].indice_dict[mod.indice_key], out.features.shape[0]))
hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, SparseConvolution)]
with torch.no_grad():
    model(torch.cat(feats), torch.cat(coords), 3)
for key in sys.argv[1:] or ["res4"]:
    mod, f, rb, n_out = [r for r in recs if r[0].indice_key == key and r[0].in_channels == r[0].out_channels][0]
    w = mod.weight.detach()
    cin = mod.in_channels
    tab, flag = rb.out_table(cin)
    for _ in range(3):
        ops.indice_conv(f, w, rb, None, n_out, False, mod.subm)
    T = 256 if key == "res4" else 512
    ncb = mod.out_channels // 64
    tr = torch.zeros(4 * ((n_out + 63) // 64) * 8, dtype=torch.int64, device=dev)
    fv2p_native.call("fv2p_sparse_conv_set_trace", tr)
    ops.indice_conv(f, w, rb, None, n_out, False, mod.subm)
    torch.cuda.synchronize()
    fv2p_native.call("fv2p_sparse_conv_set_trace", None)
    full = tr.cpu().numpy().reshape(-1, 8)
    live = full[:T * ncb]
    dur = (live[:, 3] - live[:, 2]).astype(float)
    st = tab.untyped_storage()
    allints = torch.empty(0, dtype=torch.int32, device=dev).set_(st, 0, (st.nbytes() // 4,)).cpu().numpy()
    kvol, n = tab.shape
    plan = allints[kvol * n:]
    off = 2
    lvl = 0
    while ((384 if lvl & 1 else 256) << (lvl >> 1)) != T:
        off += ((384 if lvl & 1 else 256) << (lvl >> 1)) + 1; lvl += 1
    b = plan[off:off + T + 1]
    act = tab.cpu().numpy() >= 0          # [K, n]
    bx = np.arange(T); x = bx & 7
    tile = x * (T >> 3) + np.minimum(x, T & 7) + (bx >> 3)
    rows = np.diff(b)[tile]
    pairs = np.array([act[:, b[t]:b[t + 1]].sum() for t in tile])
    groups = np.array([np.ceil(act[:, b[t]:b[t + 1]].sum(1) / 16).sum() for t in tile])
    nonempty = np.array([(act[:, b[t]:b[t + 1]].sum(1) > 0).sum() for t in tile])
    for half in range(ncb):
        d = dur[half * T:(half + 1) * T]
        X = np.stack([groups, rows, np.ones(T)], 1)
        coef = np.linalg.lstsq(X, d, rcond=None)[0]
        print(f"{key} half {half}: dur median {np.median(d):.0f} min {d.min():.0f} max {d.max():.0f}; rows {rows.min()}..{rows.max()} pairs {pairs.min()}..{pairs.max()} "
              f"groups {groups.min():.0f}..{groups.max():.0f} (mean {groups.mean():.1f}) offsets {nonempty.min()}..{nonempty.max()}; fit dur = {coef[0]:.0f}*groups + {coef[1]:.0f}*rows + {coef[2]:.0f}; "
              f"corr(dur, groups) {np.corrcoef(d, groups)[0, 1]:.2f} corr(dur, rows) {np.corrcoef(d, rows)[0, 1]:.2f} corr(dur,pairs) {np.corrcoef(d, pairs)[0, 1]:.2f}")
        lv = live[half * T:(half + 1) * T]
        nz = d > 0
        print(f"     wave 0 clocks per group: total {np.median(d[nz] / groups[nz]):.0f}  wait {np.median(lv[nz, 5] / groups[nz]):.0f}  issue {np.median(lv[nz, 6] / groups[nz]):.0f}  "
              f"compute {np.median(lv[nz, 7] / groups[nz]):.0f}  prologue {np.median(lv[nz, 4] - lv[nz, 2]):.0f} (per tile)")
        order = np.argsort(d)
        for i in list(order[:4]) + list(order[-4:]):
            print(f"   wg {i}: dur {d[i]:.0f} rows {rows[i]} pairs {pairs[i]} groups {groups[i]:.0f} offsets {nonempty[i]} xcc {live[half * T + i, 1] & 0xF}")
