"""One-process-per-GPU data parallel plumbing shared by bench.py and the gloo CPU tests.

The hot path shards by sample: every rank owns its own clouds, rulebooks and features; the only
collective is DistributedDataParallel's bucketed gradient all-reduce (RCCL over xGMI when backend == "nccl")."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def solo_ddp():
    """FV2P_DDP_SOLO=1 (measurement hook, bench.py on a one-GPU box): a ONE-rank process group over RCCL and the detector wrapped in
    DistributedDataParallel all the same, so that DDP's gradient hooks, bucket copies and RCCL's own stream run beside the step's side
    streams.  A one-rank all-reduce moves nothing; FV2P_DDP_COMM_STANDIN=<us> adds, per bucket and on the communication stream, a copy
    kernel sized to last about that long (the time a 25 MB bucket occupies a ring over xGMI: ~150 us)."""
    return os.environ.get("FV2P_DDP_SOLO") == "1"


def init_distributed(backend, device=None):
    rank, world, _ = env_world()
    if world == 1 and solo_ddp() and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # port 0: the store binds a free port itself and keeps it (no bind / close / re-use window for another process to take it)
        store = dist.TCPStore("127.0.0.1", 0, 1, True)
        dist.init_process_group(backend, store=store, rank=0, world_size=1,
                                **({"device_id": device} if (backend == "nccl" and device is not None) else {}))
        return rank, world
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        attempt = os.environ.get("FV2P_BENCH_ATTEMPT", "0")
        if attempt != "0":
            # ranks restarted by bench.py's supervisor meet at the launcher's store again: its keys of the first attempt (the dead
            # ranks' addresses) are still there, so this attempt talks under a prefix of its own
            store, rank, world = next(iter(dist.rendezvous("env://", rank=rank, world_size=world)))
            kw.update(store=dist.PrefixStore(f"fv2p_attempt_{attempt}", store), rank=rank, world_size=world)
        dist.init_process_group(backend, **kw)
    return rank, world


def rank_seeds(rank, n_pool, batch):
    """Distinct synthetic-cloud seeds per (rank, pool slot, sample): ranks never see the same cloud."""
    return [[100000 * rank + 10 * j + b for b in range(batch)] for j in range(n_pool)]


def wrap_ddp(model, device=None, find_unused_parameters=True):
    """find_unused_parameters=True is the reference trainer's setting (tools/train.py:166: its detectors have heads that
    sit out some iterations); a module that uses every parameter each step (the backbone train step of bench.py) passes
    False and saves DDP's extra autograd-graph traversal per iteration."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not solo_ddp()):
        return model
    ids = [device.index] if (device is not None and device.type == "cuda") else None
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, find_unused_parameters=find_unused_parameters,
                                                    gradient_as_bucket_view=True)
    standin_us = float(os.environ.get("FV2P_DDP_COMM_STANDIN", "0") or 0)
    if standin_us > 0 and device is not None and device.type == "cuda":
        net.register_comm_hook(_StandIn(device, standin_us), _standin_hook)
    return net


class _StandIn:
    """State of the stand-in hook: a communication stream of its own and a scratch buffer."""

    def __init__(self, device, us):
        self.stream = torch.cuda.Stream(device=device)
        self.scratch = None
        self.us = us


def _standin_hook(state, bucket):
    buf = bucket.buffer()
    fut = dist.all_reduce(buf, async_op=True).get_future()      # the real (one-rank or N-rank) RCCL call on RCCL's stream

    def after(f):
        t = f.value()[0]
        if state.scratch is None or state.scratch.numel() < t.numel():
            state.scratch = torch.empty_like(t)
        cur = torch.cuda.current_stream(t.device)
        state.stream.wait_stream(cur)
        with torch.cuda.stream(state.stream):
            # a ring all-reduce moves about twice the bucket through each GPU: two device copies of it on the communication stream
            state.scratch[:t.numel()].copy_(t)
            t.copy_(state.scratch[:t.numel()])
        cur.wait_stream(state.stream)
        return t.div_(dist.get_world_size()) if dist.get_world_size() > 1 else t
    return fut.then(after)


class FlatGradAllReduce:
    """Data parallelism with ONE collective per step: after backward the gradients are packed into one flat fp32 buffer (a multi-tensor
    copy), averaged with a single all-reduce over RCCL / xGMI, and unpacked.  The FV2P detector has 22.4 M parameters = 90 MB of
    gradients: one ring all-reduce of that moves ~160 MB through each GPU, well under a millisecond over xGMI, against a 30 ms step -
    there is nothing worth overlapping, and NOT overlapping keeps RCCL's kernels off the hardware queues while the step's own three
    streams are busy (one-rank DDP + a communication stand-in beside the backward pass: 36 ms per step; this form: 30.3,
    profiles/r05_ddp_stream_matrix.txt).  `bench.py --grad-sync flat` (the default for the FV2P workloads) / `ddp`.

    What it shares with DistributedDataParallel (tools/train.py:166 of the reference wraps its detectors in it), point by point:
      * gradients: the mean over the ranks, for every parameter some rank produced a gradient for (tests/test_dist_cpu.py holds the two
        against each other).  A parameter NO rank used keeps `grad = None` like under DDP (so AdamW neither decays it nor moves its
        moments); one that only other ranks used receives their mean.  Who used what travels in the same all-reduce (one flag per
        parameter behind the gradients); the flags are read back - a host wait - only on a step in which this rank missed a gradient.
      * start: `broadcast_parameters(src, module)` gives every rank rank `src`'s parameters AND buffers, as DDP's constructor does.
      * buffers during training (`sync_buffers()`, DDP's broadcast_buffers=True): rank 0's floating-point buffers - the BatchNorm
        running statistics - are broadcast before every forward pass as one flat tensor.  Integer buffers (num_batches_tracked) advance
        in lock step on all ranks and are only broadcast at the start.
    `timed_ms()` returns the event-timed duration of the all-reduces since the last call (bench.py prints it per step)."""

    def __init__(self, params, device=None, module=None):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n + len(self.params), dtype=torch.float32, device=self.params[0].device)   # gradients | one "used" flag per parameter
        self.flags = self.flat[n:]
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.standin = os.environ.get("FV2P_DDP_COMM_STANDIN", "0") not in ("", "0") and self.flat.is_cuda
        self.scratch = torch.empty_like(self.flat) if self.standin else None
        self.module = module
        self._fbuf = None      # floating-point buffers of the module and their flat image
        self._events = []      # (start, end) event pairs around the all-reduces, CUDA only

    def _float_buffers(self):
        if self._fbuf is None:
            bufs = [b for b in self.module.buffers() if b.is_floating_point()] if self.module is not None else []
            flat = torch.empty(sum(b.numel() for b in bufs), dtype=torch.float32, device=self.flat.device) if bufs else None
            views, off = [], 0
            for b in bufs:
                views.append(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            self._fbuf = (bufs, views, flat)
        return self._fbuf

    def broadcast_parameters(self, src=0, module=None):
        """DDP's constructor does this: every rank starts from rank `src`'s parameters and buffers."""
        if module is not None:
            self.module, self._fbuf = module, None
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src)
            if self.module is not None:
                for b in self.module.buffers():
                    dist.broadcast(b.data, src)

    @torch.no_grad()
    def sync_buffers(self, src=0):
        """DDP's broadcast_buffers=True: rank `src`'s floating-point buffers before a forward pass - pack, ONE broadcast, unpack."""
        if self.world <= 1 or self.module is None:
            return
        bufs, views, flat = self._float_buffers()
        if not bufs:
            return
        torch._foreach_copy_(views, bufs)
        dist.broadcast(flat, src)
        torch._foreach_copy_(bufs, views)

    def timed_ms(self):
        """Sum of the event-timed all-reduce durations since the last call and their number (synchronises the device)."""
        if not self._events:
            return 0.0, 0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._events)
        n = len(self._events)
        self._events = []
        return ms, n

    @torch.no_grad()
    def __call__(self):
        """Average the gradients over the ranks (call between backward and the optimiser step, on the stream the gradients were
        produced on)."""
        have = [p.grad is not None for p in self.params]
        if not all(have):
            self.flat.zero_()
        src = [p.grad for p, h in zip(self.params, have) if h]
        dst = [v for v, h in zip(self.views, have) if h]
        if src:
            torch._foreach_copy_(dst, src)
        if self.world > 1:
            if all(have):
                self.flags.fill_(1.0)
            else:
                self.flags.copy_(torch.tensor([1.0 if h else 0.0 for h in have], dtype=torch.float32), non_blocking=False)
            timed = self.flat.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(self.flat)
            if timed:
                e1.record()
                self._events.append((e0, e1))
                if len(self._events) > 4096:
                    del self._events[:2048]
            self.flat[:self.flat.numel() - len(self.params)].div_(self.world)
        if self.standin:      # measurement on one GPU: the traffic of a ring all-reduce (about twice the buffer through each GPU)
            self.scratch.copy_(self.flat)
            self.flat.copy_(self.scratch)
        if (self.world > 1 or self.standin) and src:
            torch._foreach_copy_(src, dst)          # back into the gradients: one multi-tensor launch
        if not all(have):
            # a parameter this rank has no gradient for: the other ranks' mean if any of them used it, None if nobody did (DDP leaves
            # globally unused parameters alone as well).  The flags' read is a host wait, on this rare path only.
            used = self.flags.cpu().tolist() if self.world > 1 else [0.0] * len(self.params)
            for p, v, h, u in zip(self.params, self.views, have, used):
                if not h and u > 0.0:
                    p.grad = v.clone()


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(units_per_rank_per_step, steps, seconds_max):
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return units_per_rank_per_step * world * steps / seconds_max
