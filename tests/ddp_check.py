"""Child process of tests/test_ddp_gpu.py (not collected by pytest): one rank of a two-rank DistributedDataParallel run of the FV2P
step in which BOTH ranks use GPU 0 and talk over gloo (RCCL refuses two ranks on one device; the 8-GPU node is the driver's).
    python tests/ddp_check.py <rank> <world> <port> <out dir>
Writes <out dir>/rank<r>.pt = {"single": gradients of this rank's batch without DDP, "flat": the same averaged over the ranks by
dist_utils.FlatGradAllReduce, "ddp": gradients after the first DDP backward, "losses": the three DDP steps' losses}."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.backends.cudnn.deterministic = True                    # library run-to-run noise off (tests/arrangement_check.py): the single-process
torch.use_deterministic_algorithms(True, warn_only=True)     # and the DDP pass of one rank then see bit-identical forward passes
import oracle  # noqa: E402
from fv2p_harness import dist_utils, synth  # noqa: E402
from fv2p_harness.backbone import mean_vfe  # noqa: E402
from fv2p_harness.fv2p_model import FV2PDetector, pad_gt_boxes  # noqa: E402
from test_fv2p_step_gpu import SmallFV2P  # noqa: E402

gpu = torch.device("cuda:0")
torch.cuda.set_device(gpu)
dist_utils.init_distributed("gloo", gpu)
torch.manual_seed(0)                                         # the same initial weights on every rank
cfg_streams = type("Cfg", (SmallFV2P,), {"dense_branch_stream": True, "point_branch_stream": True})     # bench.py's arrangement
cfg_inline = type("Cfg", (SmallFV2P,), {"dense_branch_stream": False, "point_branch_stream": False})
model = FV2PDetector(cfg_inline).to(gpu)


def make_inputs(seed0, batch=2, n_points=4096):
    rng = np.array(SmallFV2P.point_cloud_range, np.float32)
    clouds, boxes, feats, coords = [], [], [], []
    for b in range(batch):
        pts, bx = synth.lidar_cloud(seed0 + b, n_points, pc_range=rng, return_boxes=True)
        clouds.append(torch.from_numpy(pts).to(gpu))
        boxes.append(bx)
        v, c, k = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, rng, 5, 16000)
        feats.append(mean_vfe(torch.from_numpy(v), torch.from_numpy(k)))
        coords.append(torch.from_numpy(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1)))
    u = torch.rand(batch, SmallFV2P.nms_post + SmallFV2P.roi_per_image, generator=torch.Generator().manual_seed(1 + seed0))
    return clouds, torch.cat(feats).to(gpu), torch.cat(coords).to(gpu), pad_gt_boxes(boxes, gpu), u.to(gpu)


batches = [make_inputs(40 + 100 * rank + 10 * i) for i in range(3)]      # ranks never see the same cloud
# first step of the process on the calling stream only (MIOpen's first-call solver search: DESIGN.md 1), not part of the comparison
model(*batches[0]).backward()
torch.cuda.synchronize()
# this rank's batch without DDP
model.zero_grad(set_to_none=True)
model(*batches[0]).backward()
torch.cuda.synchronize()
single = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
# ... averaged over the ranks by the flat form (bench.py --grad-sync flat: one buffer, one all-reduce after backward)
flat_sync = dist_utils.FlatGradAllReduce([p for p in model.parameters() if p.requires_grad], gpu)
flat_sync()
torch.cuda.synchronize()
flat = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
model.zero_grad(set_to_none=True)
# the same batch under DistributedDataParallel, side streams as bench.py arranges them
model.cfg = cfg_streams
net = dist_utils.wrap_ddp(model, gpu, find_unused_parameters=False)
assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01)
losses = []
ddp = None
for i in range(3):
    loss = net(*batches[i])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    if i == 0:
        ddp = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    torch.nn.utils.clip_grad_norm_(params, SmallFV2P.grad_norm_clip)
    opt.step()
    losses.append(float(loss))
torch.cuda.synchronize()
dist_utils.barrier()
torch.save({"single": single, "ddp": ddp, "flat": flat, "losses": losses}, os.path.join(out, f"rank{rank}.pt"))
torch.distributed.destroy_process_group()
print(f"RANK {rank} DONE", losses)
