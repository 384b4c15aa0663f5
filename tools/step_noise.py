"""How far are the HIP run and the host float32 oracle run of the reduced MGAF step from the host float64 run, over several weight
seeds and input clouds?  The evidence behind K in tests/f64_calibration.py (profiles/r05_step_noise_*.txt: one file per box).
    python tools/step_noise.py [n_trials]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import f64_calibration as cal  # noqa: E402
from conftest import deterministic_libraries  # noqa: E402
from test_mgaf_head import host_runs, mgaf_group, trainable_grads, zero_gradient  # noqa: E402

gpu = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print(f"host: {os.cpu_count()} cpus, torch threads {torch.get_num_threads()}; device {torch.cuda.get_device_name(0)}")
worst, worst_param = {}, 0.0
for trial in range(n):
    model, ref, ref64, (feats, coords, gt) = host_runs(seed=trial, cloud_seed=90 + 10 * trial)
    net = model.to(gpu)
    net.taps = {}
    net.iou_peaks = ref.taps["terms"]["_iou_peaks"].to(gpu)
    with deterministic_libraries():
        net(feats.to(gpu), coords.to(gpu), 2, gt.to(gpu)).backward()
    gh, g32, g64 = trainable_grads(net), trainable_grads(ref), trainable_grads(ref64)
    rows, bad = cal.compare(gh, g32, g64, mgaf_group, zero_gradient)
    print(cal.report(rows, f"trial {trial} (weights seed {trial}, clouds {90 + 10 * trial}+): {'OK' if not bad else 'FAIL ' + '; '.join(bad)}"))
    worst_param = max(worst_param, max(cal.distances(gh, g64, zero_gradient).values()))
    for g, _, hmax, rmax, hmed, rmed, hpool, rpool in rows:
        w = worst.get(g, (0.0, 0.0, 0.0))
        worst[g] = (max(w[0], hmed / rmed), max(w[1], hpool / rpool), max(w[2], hmax / rmax))
print("largest ratio hip / host32 over the trials (median, pooled, max statistic):", {g: tuple(round(x, 2) for x in v) for g, v in worst.items()})
print(f"worst single HIP parameter over the trials: {worst_param:.2e} from float64")
