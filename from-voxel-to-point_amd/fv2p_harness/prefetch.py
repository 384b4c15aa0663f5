"""Input pipeline thread: prepares batch t+1 (GPU voxelisation, MeanVFE, optionally the rulebooks) on its own HIP stream
while batch t trains — the role the reference gives to its DataLoader workers, which voxelise on the CPU in parallel
with the training step (pcdet/datasets/dataset.py:122-150, tools/train_utils/train_utils.py:20-27).

The library calls release the GIL (ctypes), so the dozen kernel launches and the host synchronisations inside a
voxelisation / rulebook build proceed while the main thread issues the training step."""
import queue
import threading

import torch


class BatchPrefetcher(object):
    def __init__(self, produce, device, depth=1):
        """produce(i) -> tuple of tensors (any nesting of tuples/lists/dicts of tensors is walked for record_stream)."""
        self.produce, self.device = produce, device
        self.stream = torch.cuda.Stream(device=device)
        self.jobs, self.done = queue.Queue(), queue.Queue()
        self.depth, self.pending = depth, 0
        self.retire = queue.Queue()   # (batch, event on the training stream after its last use)
        self.last = None
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
                    out = self.produce(i)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                self.done.put((out, ev, None))
            except BaseException as e:  # surfaced by get()
                self.done.put((None, None, e))

    def submit(self, i):
        self.jobs.put(i)
        self.pending += 1

    def get(self):
        """Next prepared batch.  Contract: a batch (and the rulebooks attached to it) is used only until the next get()
        — its memory belongs to the pipeline stream and is recycled once the training stream passes that point."""
        out, ev, err = self.done.get()
        self.pending -= 1
        if err is not None:
            raise err
        main = torch.cuda.current_stream(self.device)
        if self.last is not None:
            # the previous batch must not be touched after this call: everything that used it is already on `main`
            done_ev = torch.cuda.Event()
            done_ev.record(main)
            self.retire.put((self.last, done_ev))
        main.wait_event(ev)
        self.last = out
        return out

    def close(self):
        self.jobs.put(None)
        self.thread.join(timeout=5)


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
