"""Host-side cost of one forward / backward of the bench step (cProfile, cumulative).  python tools/hostprof.py [fwd|bwd]"""
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "from-voxel-to-point_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

sys.argv = ["bench.py", "--cpu-clouds", "0"]
args = bench.parse()
dev = torch.device("cuda:0")
model, step, voxelize, pool = bench.build_step(args, dev, 0, 1)
for i in range(5):
    step(i)
torch.cuda.synchronize()
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
inputs = [voxelize(pool[i % 4]) for i in range(4)]
# rulebooks prefetched as in the benchmark's input pipeline: the forward below only issues conv / BatchNorm work
from pcdet.ops import spconv  # noqa: E402
with torch.no_grad():
    recipe = spconv.rulebook_recipe(model(inputs[0][0], inputs[0][1], args.batch)[0].indice_dict, inputs[0][1])
for f, c in inputs:
    spconv.attach_rulebooks(c, spconv.build_rulebooks(recipe, c, args.batch))
torch.cuda.synchronize()
# plain timing first (no profiler): host time of the forward alone
t0 = time.perf_counter()
outs = []
for i in range(20):
    f, c = inputs[i % 4]
    outs.append(model(f, c, args.batch)[0].features.square().mean())
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"forward host issue {(t1 - t0) / 20 * 1e3:.3f} ms (wall incl. drain {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms)")
prof = cProfile.Profile()
prof.enable()
for i in range(20):
    f, c = inputs[i % 4]
    loss = model(f, c, args.batch)[0].features.square().mean()
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
