"""Host -> device feed of the training loops (new: the reference moves every batch with a blocking
`.to(device)` from pageable memory inside the step, /root/reference/gridnext/training.py:47-51, :135-139).

`DevicePrefetcher(loader, device)` wraps any iterable of `(inputs, labels)` batches (a `DataLoader`): a producer thread
pulls the next batch, stages it in PINNED host buffers (a small ring, reused: page-locking 245 MB per array every step
would cost more than the copy), issues the host -> device copies `non_blocking` on a side HIP stream and records an
event; the consumer (the training loop) makes its own stream wait for that event and gets device tensors.  Batch i + 1
is therefore on its way over PCIe while step i computes; nothing in the step synchronises with the host.

Patches should travel as uint8 (`PatchGridDataset(..., raw_uint8=True)`): a 128-px Visium array is 245 MB instead of
981 MB, ToTensor's / 255 happens inside the stem kernel (csrc/stem_pool.hip), and the host side never touches floats.
Batches that are already on the device pass through untouched.
"""
import queue
import threading

import torch

_STOP = object()


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, fn) for o in obj)
    return obj


def _tensors(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _tensors(o, out)
    return out


class _PinnedRing:
    """depth + 1 slots of pinned staging buffers, keyed by (position in the batch, shape, dtype) and grown on demand."""

    def __init__(self, slots):
        self.slots = [dict() for _ in range(slots)]
        self.events = [None] * slots

    def stage(self, slot, index, t):
        key = (index, tuple(t.shape), t.dtype)
        buf = self.slots[slot].get(key)
        if buf is None:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self.slots[slot][key] = buf
        buf.copy_(t)
        return buf


class DevicePrefetcher:
    """Iterate `loader` one batch ahead of the consumer, delivering device tensors.

        for inputs, labels in DevicePrefetcher(loader, device):
            ...

    depth: batches in flight ahead of the consumer (2 = double buffering)."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.bytes_moved = 0

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        if self.device.type != 'cuda':
            yield from self.loader
            return
        # device-resident data (tensors already in HBM: benchmarks, small data sets): no thread, no copies
        it = iter(self.loader)
        try:
            first = next(it)
        except StopIteration:
            return
        if all(t.is_cuda for t in _tensors(first, [])):
            yield first
            yield from it
            return

        def batches():
            yield first
            yield from it

        q = queue.Queue(maxsize=self.depth)
        ring = _PinnedRing(self.depth + 2)
        side = torch.cuda.Stream(device=self.device)
        failure = []
        stop = threading.Event()

        def produce():
            try:
                torch.cuda.set_device(self.device)
                for n, batch in enumerate(batches()):
                    if stop.is_set():
                        break
                    slot = n % len(ring.slots)
                    if ring.events[slot] is not None:
                        ring.events[slot].synchronize()          # the copies that last read this slot's buffers are done
                    flat = _tensors(batch, [])
                    if all(t.is_cuda for t in flat):
                        q.put((batch, None))                     # already resident: nothing to move
                        continue
                    counter = [0]

                    def move(t):
                        i = counter[0]
                        counter[0] += 1
                        if t.is_cuda:
                            return t
                        src = t if t.is_pinned() else ring.stage(slot, i, t)
                        self.bytes_moved += src.numel() * src.element_size()
                        with torch.cuda.stream(side):
                            return src.to(self.device, non_blocking=True)

                    dev_batch = _map(batch, move)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    ring.events[slot] = ev
                    q.put((dev_batch, ev))
            except BaseException as exc:                         # surfaced in the consumer thread
                failure.append(exc)
            finally:
                q.put(_STOP)

        thread = threading.Thread(target=produce, name='gnx-prefetch', daemon=True)
        thread.start()
        try:
            while True:
                item = q.get()
                if item is _STOP:
                    break
                batch, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)                           # device-side ordering only: the host does not block
                    for t in _tensors(batch, []):
                        t.record_stream(cur)                     # allocated on the side stream, consumed on this one
                yield batch
        finally:
            # the consumer stopped (normally or early): let the producer finish its current batch and leave
            stop.set()
            while thread.is_alive():
                try:
                    if q.get(timeout=0.05) is _STOP:
                        break
                except queue.Empty:
                    pass
            thread.join(timeout=5.0)
        if failure:
            raise failure[0]


def wrap(loader, device):
    """The loops' feed: a DevicePrefetcher on a HIP device, the loader itself otherwise (or when it already is one)."""
    if isinstance(loader, DevicePrefetcher) or torch.device(device).type != 'cuda':
        return loader
    return DevicePrefetcher(loader, device)
