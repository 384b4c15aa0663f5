"""Input pipeline thread: prepares batch t+1 (GPU voxelisation, MeanVFE, optionally the rulebooks) on its own HIP stream
while batch t trains — the role the reference gives to its DataLoader workers, which voxelise on the CPU in parallel
with the training step (pcdet/datasets/dataset.py:122-150, tools/train_utils/train_utils.py:20-27).

The library calls release the GIL (ctypes), so the dozen kernel launches and the host synchronisations inside a
voxelisation / rulebook build proceed while the main thread issues the training step."""
import queue
import threading

import torch


class _Worker(object):
    def __init__(self, owner, device):
        self.owner, self.device = owner, device
        self.stream = torch.cuda.Stream(device=device)
        self.jobs, self.done, self.retire = queue.Queue(), queue.Queue(), queue.Queue()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            i = self.jobs.get()
            if i is None:
                return
            # batches the training stream has finished with: order this stream after their last use, then let go of
            # them — their blocks return to this stream's pool in stream order (no record_stream bookkeeping, which
            # defers every free behind an event and keeps the pool growing)
            while True:
                try:
                    old, ev = self.retire.get_nowait()
                except queue.Empty:
                    break
                self.stream.wait_event(ev)
                del old
            try:
                with torch.cuda.stream(self.stream):
                    out = self.owner.produce(i)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                self.done.put((out, ev, None))
            except BaseException as e:  # surfaced by get()
                self.done.put((None, None, e))


class BatchPrefetcher(object):
    def __init__(self, produce, device, workers=1):
        """produce(i) -> the batch (tensors, possibly nested / with rulebooks attached).  `workers` pipeline threads, each
        with its own HIP stream, prepare consecutive batches concurrently (a batch's preparation is a chain of small
        kernels and host synchronisations — latency, not throughput — so two in flight nearly double the rate);
        batches are handed out in submission order."""
        self.produce, self.device = produce, device
        self.workers = [_Worker(self, device) for _ in range(max(1, int(workers)))]
        self.pending = 0
        self.n_in = self.n_out = 0
        self.last = None   # (batch, worker) handed out by the previous get()

    def submit(self, i):
        self.workers[self.n_in % len(self.workers)].jobs.put(i)
        self.n_in += 1
        self.pending += 1

    def get(self):
        """Next prepared batch.  Contract: a batch (and the rulebooks attached to it) is used only until the next get()
        — its memory belongs to its pipeline stream and is recycled once the training stream passes that point."""
        w = self.workers[self.n_out % len(self.workers)]
        out, ev, err = w.done.get()
        self.n_out += 1
        self.pending -= 1
        if err is not None:
            raise err
        main = torch.cuda.current_stream(self.device)
        if self.last is not None:
            # the previous batch must not be touched after this call: everything that used it is already on `main`
            done_ev = torch.cuda.Event()
            done_ev.record(main)
            self.last[1].retire.put((self.last[0], done_ev))
        main.wait_event(ev)
        self.last = (out, w)
        return out

    def close(self):
        for w in self.workers:
            w.jobs.put(None)
        for w in self.workers:
            w.thread.join(timeout=5)


class BatchAhead(object):
    """The same hand-over without a thread: prepare(i) enqueues batch i on a side stream from the training thread itself —
    called between the forward and the backward pass of the step before — and take(i) orders the training stream after it.
    The preparation's host synchronisations (voxel counts, rulebook sizes) then wait for the side stream only, whose work is
    a few hundred microseconds, instead of for everything the training stream still has queued: the host can run a whole
    step ahead of the device.  Pays when the step is device bound; a thread pays when it is launch bound."""

    def __init__(self, produce, device, priority=0, stream=None):
        self.produce, self.device = produce, device
        # `stream`: an existing side stream to share (a process should not drive more side streams than it needs: four hardware queues)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=device, priority=priority)
        self.ready = {}       # i -> (batch, event on the side stream)
        self.current = None   # batch handed out by the last take()
        self.retired = []     # (batch, event on the training stream after which nothing touches it, step that recorded it)

    def prepare(self, i):
        if i in self.ready:
            return
        # blocks of a finished batch return to the side stream's pool in stream order (see _Worker._run).  Only batches retired
        # a full step ago are let go here: the side stream must not wait for an event the device has yet to reach, or the
        # preparation's host synchronisations would wait for the training stream after all.
        while self.retired and self.retired[0][2] <= i - 2:
            old, ev, _ = self.retired.pop(0)
            self.stream.wait_event(ev)
            del old
        with torch.cuda.stream(self.stream):
            out = self.produce(i)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.ready[i] = (out, ev)

    def take(self, i):
        """Batch i (prepared now if prepare(i) was not called).  Contract as BatchPrefetcher.get: use it until the next take()."""
        self.prepare(i)
        out, ev = self.ready.pop(i)
        main = torch.cuda.current_stream(self.device)
        if self.current is not None:
            done = torch.cuda.Event()
            done.record(main)
            self.retired.append((self.current, done, i))
        main.wait_event(ev)
        self.current = out
        return out


def _walk(obj, fn, seen=None):
    """Applies fn to every CUDA tensor reachable from obj (containers, object attributes, attached rulebooks)."""
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            fn(obj)
        extra = getattr(obj, "_fv2p_indice_dict", None)   # rulebooks attached by spconv.attach_rulebooks
        if extra:
            _walk(extra, fn, seen)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _walk(o, fn, seen)
    elif isinstance(obj, dict):
        for o in obj.values():
            _walk(o, fn, seen)
    elif hasattr(obj, "__dict__"):
        for o in vars(obj).values():
            _walk(o, fn, seen)
