"""The multi-rank path on a GPU: two DistributedDataParallel ranks of the FV2P step, both on GPU 0, gloo between them (RCCL does
not take two ranks on one device; bench.py has the same hooks, FV2P_FORCE_DEVICE / FV2P_DIST_BACKEND).  What this covers that
tests/test_dist_cpu.py cannot: DDP's gradient hooks and bucket copies running beside the step's own side streams (dense branch, point
branch, key-point sampling, weight gradients) and the deferred weight-gradient gate node (tools/train.py:166 is the reference site)."""
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_fv2p_ddp(gpu, tmp_path):
    """Both gradient-averaging forms of bench.py on two ranks: the flat all-reduce after backward gives both ranks exactly the mean of
    their single-process gradients.  Then three DDP steps with bench.py's stream arrangement, no hang within the time limit; after the first backward both ranks hold the
    same gradients and they are the MEAN of the two ranks' single-process gradients (per parameter: 2e-5 of its norm - the repo's
    float-atomic interpolation gradient alone moves a gradient by 3e-6 between two runs; the forward passes are bit-identical under
    the deterministic library settings of the child)."""
    port = str(29600 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_check.py"), str(r), "2", port, str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420)[0])
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("two DDP ranks on one GPU did not finish three steps within 420 s: a hang of the multi-rank path is a defect")
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK {r} DONE" in o, o[-3000:]
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2))
    assert all(torch.isfinite(torch.tensor(r["losses"])).all() for r in (r0, r1))
    from test_fv2p_step_gpu import zero_gradient
    worst = ("", 0.0)
    for k, g0 in r0["ddp"].items():
        assert torch.equal(g0, r1["ddp"][k]), f"{k}: the ranks hold different gradients after the all-reduce"
        if zero_gradient(k):      # a conv bias in front of train-mode BatchNorm: analytically zero, rounding noise on both sides
            continue
        mean = (r0["single"][k].double() + r1["single"][k].double()) / 2
        err = float((g0.double() - mean).norm())
        bound = 2e-5 * float(mean.norm()) + 1e-7 * mean.numel() ** 0.5
        worst = max(worst, (k, err / max(float(mean.norm()), 1e-30)), key=lambda t: t[1])
        assert err < bound, f"{k}: all-reduced gradient differs from the mean of the single-process gradients by {err:.3e} (bound {bound:.3e})"
    print("worst parameter:", worst)
    # the flat form (one buffer, ONE all-reduce after backward: bench.py's --grad-sync flat) gives every rank the same mean
    for k, g0 in r0["flat"].items():
        assert torch.equal(g0, r1["flat"][k]), f"{k}: the ranks hold different gradients after the flat all-reduce"
        mean = (r0["single"][k].double() + r1["single"][k].double()) / 2
        assert float((g0.double() - mean).norm()) <= 1e-6 * float(mean.norm()) + 1e-9 * mean.numel() ** 0.5, k
