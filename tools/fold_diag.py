"""Diagnostic: the reduced FV2P step's gradients under the fold arrangement - in line, with side streams, under a one-rank DDP."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29871")
import numpy as np, torch
torch.backends.cudnn.deterministic = True
torch.use_deterministic_algorithms(True, warn_only=True)
import oracle
from fv2p_harness import dist_utils, synth
from fv2p_harness.backbone import mean_vfe
from fv2p_harness.fv2p_model import FV2PDetector, pad_gt_boxes
from test_fv2p_step_gpu import SmallFV2P
import pcdet.ops.spconv as spconv
gpu = torch.device("cuda:0"); torch.cuda.set_device(gpu)
torch.distributed.init_process_group("gloo", rank=0, world_size=1)
torch.manual_seed(0)
cfg_streams = type("Cfg", (SmallFV2P,), {"dense_branch_stream": True, "point_branch_stream": True})
cfg_inline = type("Cfg", (SmallFV2P,), {"dense_branch_stream": False, "point_branch_stream": False})
model = FV2PDetector(cfg_inline).to(gpu)
def make_inputs(seed0, batch=2, n_points=4096):
    rng = np.array(SmallFV2P.point_cloud_range, np.float32)
    clouds, boxes, feats, coords = [], [], [], []
    for b in range(batch):
        pts, bx = synth.lidar_cloud(seed0 + b, n_points, pc_range=rng, return_boxes=True)
        clouds.append(torch.from_numpy(pts).to(gpu)); boxes.append(bx)
        v, c, k = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, rng, 5, 16000)
        feats.append(mean_vfe(torch.from_numpy(v), torch.from_numpy(k)))
        coords.append(torch.from_numpy(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1)))
    u = torch.rand(batch, SmallFV2P.nms_post + SmallFV2P.roi_per_image, generator=torch.Generator().manual_seed(1 + seed0))
    return clouds, torch.cat(feats).to(gpu), torch.cat(coords).to(gpu), pad_gt_boxes(boxes, gpu), u.to(gpu)
batch = make_inputs(40)
state = {k: v.clone() for k, v in model.state_dict().items()}
def grads(cfg, fold, net=None):
    spconv.set_bn_fold(fold)
    model.load_state_dict(state)
    model.cfg = cfg
    model.zero_grad(set_to_none=True)
    loss = (net or model)(*batch)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
for _ in range(2): grads(cfg_inline, True)
ref = grads(cfg_inline, False)
def cmp(name, got):
    worst = max(((float((got[1][k].double() - ref[1][k].double()).norm() / max(float(ref[1][k].double().norm()), 1e-30)), k) for k in ref[1] if "conv1.bias" not in k and "conv2.bias" not in k), key=lambda t: t[0])
    print(f"{name:40s} loss {got[0]:.6f} (ref {ref[0]:.6f})  worst rel {worst[0]:.2e} at {worst[1]}", flush=True)
cmp("inline fold", grads(cfg_inline, True))
cmp("inline fold again", grads(cfg_inline, True))
cmp("streams nofold", grads(cfg_streams, False))
cmp("streams fold", grads(cfg_streams, True))
cmp("streams fold again", grads(cfg_streams, True))
net = dist_utils.wrap_ddp(model, gpu, find_unused_parameters=False) if False else torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False, gradient_as_bucket_view=True)
cmp("ddp inline fold", grads(cfg_inline, True, net))
cmp("ddp streams fold", grads(cfg_streams, True, net))
cmp("ddp streams nofold", grads(cfg_streams, False, net))
cmp("ddp streams fold again", grads(cfg_streams, True, net))
