"""Host-side cost of the boundary leg (bench.py's step.inline(i, boundary=True)): wall per step, then cProfile by own time and cumulative.
python tools/boundary_prof.py [steps]"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sys.argv = ["bench.py", "--cpu-clouds", "0"]
args = bench.parse()
dev = torch.device("cuda:0")
model, step, voxelize, pool = bench.build_fv2p_step(args, dev, 0, 1)
for i in range(5):
    step.inline(i, boundary=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step.inline(i, boundary=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"boundary leg: host issue {(t1 - t0) / steps * 1e3:.2f} ms per step, wall incl. drain {(t2 - t0) / steps * 1e3:.2f} ms")
prof = cProfile.Profile()
prof.enable()
for i in range(steps):
    step.inline(i, boundary=True)
prof.disable()
torch.cuda.synchronize()
for key, n in (("tottime", 45), ("cumulative", 70)):
    print(f"==== by {key} (over {steps} steps)")
    pstats.Stats(prof).sort_stats(key).print_stats(n)
