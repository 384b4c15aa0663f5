"""CPU, world_size 2, gloo: the N>1 plumbing of bench.py (per-rank data, barrier, max-over-ranks timing, aggregate
throughput) and DistributedDataParallel over a network built from pcdet.ops.spconv modules (run here through the
CPU oracle mirror, since the HIP path needs a GPU): after backward every rank holds the mean of the per-rank gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    for p in (REPO, os.path.join(REPO, "from-voxel-to-point_amd"), os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import numpy as np
    from fv2p_harness import dist_utils
    from oracle.spconv_cpu import cpu_mirror
    from pcdet.ops import spconv
    from sparse_util import random_active
    r, w = dist_utils.init_distributed("gloo")
    assert (r, w) == (rank, world)
    seeds = dist_utils.rank_seeds(rank, 2, 3)
    torch.manual_seed(0)  # identical initial weights on every rank (DDP also broadcasts rank 0's)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(4, 8, 3, padding=1, bias=False, indice_key="s1"), torch.nn.BatchNorm1d(8), torch.nn.ReLU(),
        spconv.SparseConv3d(8, 16, 3, stride=2, padding=1, bias=True, indice_key="d1"))
    ref = cpu_mirror(net)

    class Step(torch.nn.Module):
        """DDP needs tensors (not SparseConvTensor objects) in the forward output to trace used parameters; the
        reference's DDP-wrapped detectors likewise return loss tensors (tools/train_utils/train_utils.py:27)."""

        def __init__(self, body):
            super().__init__()
            self.body = body

        def forward(self, feats, coords, shape, batch):
            return self.body(spconv.SparseConvTensor(feats, coords, shape, batch)).features.square().mean()

    ddp = dist_utils.wrap_ddp(Step(ref))
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    ind = random_active(seeds[0][0], 2, [6, 10, 10], 150)
    feats = torch.from_numpy(np.random.default_rng(seeds[0][1]).standard_normal((ind.shape[0], 4)).astype(np.float32))
    ddp(feats, torch.from_numpy(ind), [6, 10, 10], 2).backward()
    grads = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    gathered = [torch.zeros_like(grads) for _ in range(world)]
    dist.all_gather(gathered, grads)
    dist_utils.barrier()
    dt = dist_utils.max_over_ranks(1.0 + rank)
    thr = dist_utils.aggregate_throughput(4, 10, dt)
    if rank == 0:
        torch.save({"seeds": seeds, "same": bool(torch.equal(gathered[0], gathered[1])), "dt": dt, "thr": thr,
                    "gnorm": float(grads.norm())}, out)
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["same"] and res["gnorm"] > 0            # all-reduced gradients are identical on both ranks
    assert res["dt"] == 2.0                              # MAX over ranks (rank 1 reported 2.0 s)
    assert res["thr"] == pytest.approx(4 * 2 * 10 / 2.0)  # whole-job units / max time
    assert res["seeds"][0][0] == 0                       # rank 0's first seed; rank 1 starts at 100000


def test_rank_seeds_are_disjoint():
    sys.path.insert(0, os.path.join(REPO, "from-voxel-to-point_amd"))
    from fv2p_harness import dist_utils
    a = {s for row in dist_utils.rank_seeds(0, 4, 4) for s in row}
    b = {s for row in dist_utils.rank_seeds(1, 4, 4) for s in row}
    assert not (a & b) and len(a) == 16


def test_core_blocks_are_disjoint_per_local_rank(monkeypatch):
    """bench.pin_cores: every local rank takes its own compact block out of the affinity mask it inherited; small masks and
    platforms without sched_setaffinity are left alone."""
    sys.path.insert(0, REPO)
    import bench
    mask = set(range(256))
    got = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(mask), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: got.__setitem__("cores", list(cores)), raising=False)
    blocks = [bench.pin_cores(r, 8, 16) for r in range(8)]
    assert all(len(b) == 16 for b in blocks) and len(set().union(*map(set, blocks))) == 128
    assert blocks[0] == list(range(16)) and blocks[4] == list(range(64, 80))      # ranks 0-3 / 4-7 on the two sockets' first cores
    assert got["cores"] == blocks[7]
    assert bench.pin_cores(0, 1, 0) is None                                        # switched off
    mask = set(range(8))
    assert bench.pin_cores(0, 1, 16) is None                                       # launcher already restricted us to 8 cores
    mask = set(range(256))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: (_ for _ in ()).throw(OSError("not permitted")), raising=False)
    assert bench.pin_cores(0, 1, 16) is None


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks as a child torchrun and relays
    rank 0's line (tools/scripts/dist_train.sh:26 in the reference); --dry-run keeps the GPU out of it."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line)["n_gpus"] == 2


def test_bench_watchdog_repeats_a_hung_run_in_line():
    """A measurement that produces nothing within --watchdog seconds is stopped (its own process group) and repeated once with
    the side-stream input pipelines off; FV2P_BENCH_TEST_HANG makes the first attempt sleep forever."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FV2P_BENCH_INNER")}
    env["FV2P_BENCH_TEST_HANG"] = "1"
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run", "--watchdog", "60", "--stall", "5"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "repeating it with --fps-ahead 0" in out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line)["n_gpus"] == 2


def test_ranks_of_an_outside_launcher_supervise_themselves():
    """Under somebody else's launcher (RANK set) every rank process stays a supervisor and runs the real rank as its child: when the
    ranks stop stepping — here with their process group and rank 0's store alive — each supervisor stops its child and starts it
    again with the side-stream arrangements off; the second set of ranks meets at the same master port and reports."""
    import json
    import socket
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FV2P_BENCH_INNER")}
    env["FV2P_BENCH_TEST_HANG"] = "2"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           bench, "--gpus", "2", "--dry-run", "--watchdog", "60", "--stall", "5"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stderr.count("repeating it with --fps-ahead 0") == 2
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line)["n_gpus"] == 2


def test_helper_thread_runs_one_job_and_surfaces_its_error():
    """bench.Later (the helper thread that prepares the next batch beside the backward pass): run() starts a job, join() waits for it,
    a second run() waits for the first, and what a job raises comes out of the next join()."""
    import threading
    import bench
    later, seen, gate = bench.Later(), [], threading.Event()
    later.join()                                            # nothing pending: no-op
    later.run(lambda: (gate.wait(5), seen.append("a")))
    assert seen == []
    gate.set()
    later.run(seen.append, "b")                             # joins the first job before starting the second
    later.join()
    assert seen == ["a", "b"]

    def boom():
        raise ValueError("from the helper")
    later.run(boom)
    with pytest.raises(ValueError, match="from the helper"):
        later.join()
    later.join()                                            # the error is reported once


def test_lean_gradient_clipping_is_torchs_bit_for_bit():
    """fv2p_harness.optim.clip_grad_norm_ (the bench step's GRAD_NORM_CLIP, train_utils.py:43): the same foreach kernels in the same
    order as torch.nn.utils.clip_grad_norm_ — total norm and clipped gradients bit for bit, clipping active and inactive, parameters
    without a gradient skipped."""
    from fv2p_harness.optim import clip_grad_norm_
    torch.manual_seed(0)
    for scale, max_norm in ((3.0, 2.0), (0.01, 10.0)):
        ps = [torch.nn.Parameter(torch.randn(s)) for s in [(3, 4), (10,), (5, 5, 2), (7,)]]
        for p in ps[:3]:
            p.grad = torch.randn_like(p) * scale
        qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        for p, q in zip(ps[:3], qs[:3]):
            q.grad = p.grad.clone()
        a = clip_grad_norm_(ps, max_norm)
        b = torch.nn.utils.clip_grad_norm_(qs, max_norm, foreach=True)
        assert torch.equal(a, b) and all(torch.equal(p.grad, q.grad) for p, q in zip(ps[:3], qs[:3])) and ps[3].grad is None


def _flat_worker(rank, world, port, out):
    for p in (REPO, os.path.join(REPO, "from-voxel-to-point_amd"), os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import copy
    from fv2p_harness import dist_utils
    dist_utils.init_distributed("gloo")
    torch.manual_seed(1 + rank)                     # DIFFERENT initial weights per rank: broadcast_parameters must make them rank 0's
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    unused = torch.nn.Linear(4, 4)                  # a module rank 1 never runs: its parameters have no gradient there
    never = torch.nn.Linear(3, 3)                   # a module NO rank runs: its gradients must stay None (DDP leaves them alone too)
    bn = torch.nn.BatchNorm1d(6, momentum=0.5)      # buffers: running statistics + num_batches_tracked
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 1))
    holder = torch.nn.ModuleList([net, unused, never, bn])
    params = list(net.parameters()) + list(unused.parameters())
    sync = dist_utils.FlatGradAllReduce(params + list(never.parameters()), module=holder)
    sync.broadcast_parameters(0)
    buf0 = bn.running_mean.clone()                  # rank 0's value (1.0) on every rank after the constructor-style broadcast
    bn.train()
    bn(torch.randn(8, 6, generator=torch.Generator().manual_seed(50 + rank)))     # per-rank statistics: the buffers drift apart
    drift = bn.running_mean.clone()
    sync.sync_buffers(0)                            # DDP's broadcast_buffers=True: rank 0's again
    bufs = [torch.zeros_like(drift) for _ in range(world)]
    dist.all_gather(bufs, bn.running_mean.clone())
    drifts = [torch.zeros_like(drift) for _ in range(world)]
    dist.all_gather(drifts, drift)
    w0 = torch.cat([p.detach().reshape(-1) for p in params])
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    ddp_net = torch.nn.parallel.DistributedDataParallel(copy.deepcopy(net))
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 + rank))      # every rank its own batch
    loss = net(x).square().mean()
    if rank == 0:
        loss = loss + unused(x[:, :4]).square().mean()
    loss.backward()
    local = [None if p.grad is None else p.grad.clone() for p in params]
    sync()
    ddp_net(x).square().mean().backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    ddp = torch.cat([p.grad.reshape(-1) for p in ddp_net.parameters()])
    g_unused = torch.cat([p.grad.reshape(-1) for p in unused.parameters()])
    both = [torch.zeros_like(g_unused) for _ in range(world)]
    dist.all_gather(both, g_unused)
    loc = torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for p, g in zip(unused.parameters(), local[-2:])])
    locs = [torch.zeros_like(loc) for _ in range(world)]
    dist.all_gather(locs, loc)
    if rank == 0:
        torch.save({"weights_equal": bool(torch.equal(gathered[0], gathered[1])), "flat_vs_ddp": float((flat - ddp).abs().max()),
                    "scale": float(ddp.abs().max()), "unused_same": bool(torch.equal(both[0], both[1])),
                    "unused_is_mean": float((both[0] - (locs[0] + locs[1]) / 2).abs().max()), "unused_norm": float(both[0].norm()),
                    "never_none": all(p.grad is None for p in never.parameters()), "buf_start": float(buf0.mean()),
                    "buf_drifted": not torch.equal(drifts[0], drifts[1]), "buf_synced": bool(torch.equal(bufs[0], bufs[1]) and torch.equal(bufs[0], drifts[0]))}, out)
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_equals_ddp(tmp_path):
    """fv2p_harness.dist_utils.FlatGradAllReduce (bench.py --grad-sync flat: one flat buffer, ONE all-reduce after backward) on two gloo
    ranks: after broadcast_parameters every rank holds rank 0's weights; after the call every rank holds the gradients
    DistributedDataParallel computes for the same batches (mean over the ranks, 1e-6); a parameter that has a gradient on one rank only
    gets the mean with zeros for the other - DDP's find_unused_parameters behaviour - and the same value on both ranks; a parameter no
    rank used keeps grad None; module buffers are rank 0's after broadcast_parameters and after every sync_buffers()."""
    out = str(tmp_path / "flat.pt")
    port = 29500 + (os.getpid() * 7) % 400
    mp.spawn(_flat_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["weights_equal"]
    assert r["flat_vs_ddp"] <= 1e-6 * max(r["scale"], 1.0), r
    assert r["unused_same"] and r["unused_is_mean"] <= 1e-7 and r["unused_norm"] > 0, r
    # round 6: a parameter no rank used keeps grad None; buffers follow rank 0 at the start and at every sync_buffers()
    assert r["never_none"], r
    assert r["buf_start"] == 1.0 and r["buf_drifted"] and r["buf_synced"], r
