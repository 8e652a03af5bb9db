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

    def buffer(self, slot, index, shape, dtype):
        key = (index, tuple(shape), dtype)
        buf = self.slots[slot].get(key)
        if buf is None:
            try:
                buf = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
            except RuntimeError:                              # the host refuses to page-lock more: pageable staging
                buf = torch.empty(tuple(shape), dtype=dtype)  # (the copy out of it is then synchronous, still correct)
            self.slots[slot][key] = buf
        return buf

    def stage(self, slot, index, t):
        buf = self.buffer(slot, index, t.shape, t.dtype)
        buf.copy_(t)
        return buf

    def prime(self, slot):
        """Give every other slot the buffers `slot` has.  Page-locking 285 MB takes ~0.1 s on some hosts: done lazily, slots
        2 and 3 were locked while steps 2 and 3 were already waiting for their batches (measured as 64 -> 80 ms per step
        over the first steps on a slow box); done here it all happens before the first batch is delivered."""
        for other, table in enumerate(self.slots):
            if other != slot:
                for (index, shape, dtype) in list(self.slots[slot]):
                    self.buffer(other, index, shape, dtype)


def _default_collate_loader(loader):
    """True for a DataLoader whose batches this module can assemble itself: automatic batching, torch's default collate,
    no worker processes.  Then samples are stacked DIRECTLY into the pinned staging buffers - default_collate would first
    stack them into a fresh pageable tensor (measured on the GPU box: 61 ms for one 245 MB uint8 array, nearly all of it
    page faults of the new allocation; the copy into an existing pinned buffer takes < 1 ms)."""
    try:
        from torch.utils.data import DataLoader, IterableDataset
        from torch.utils.data._utils.collate import default_collate
    except ImportError:                                       # pragma: no cover
        return False
    # map-style datasets only: torch gives a loader over an IterableDataset a BatchSampler over an ENDLESS sampler
    # (`_InfiniteConstantSampler`), so `batch_sampler is not None` holds there too and drawing "one epoch" of index batches
    # from it would never return.  Such loaders are iterated as they are.
    return (isinstance(loader, DataLoader) and loader.num_workers == 0 and loader.batch_sampler is not None
            and not isinstance(loader.dataset, IterableDataset)
            and loader.collate_fn is default_collate and not getattr(loader, 'pin_memory', False))


def _epoch_index_batches(loader):
    """The index batches of one epoch of `loader`, drawn as its own iterator would draw them: a DataLoader iterator first takes
    one number from the loader's generator (its worker base seed, torch/utils/data/dataloader.py `_BaseDataLoaderIter`), then
    the sampler permutes - skipping that draw would give another order than the plain loader for the same seed.  Drawn
    eagerly in the calling (training) thread, so the random-number stream is consumed at the same point of the program."""
    it = iter(loader.batch_sampler)
    torch.empty((), dtype=torch.int64).random_(generator=loader.generator)
    return [list(b) for b in it]


def _collate_pinned(ring, slot, samples, counter):
    """default_collate's structure (tensors stacked along a new batch dimension, sequences collated element-wise into
    LISTS, numbers to tensors) with every tensor stacked straight into a pinned buffer of ring slot `slot`."""
    first = samples[0]
    if torch.is_tensor(first):
        i = counter[0]
        counter[0] += 1
        if first.is_cuda:
            return torch.stack(samples, 0)
        out = ring.buffer(slot, i, (len(samples),) + tuple(first.shape), first.dtype)
        return torch.stack(samples, 0, out=out)
    if isinstance(first, (list, tuple)):
        return [_collate_pinned(ring, slot, [smp[k] for smp in samples], counter) for k in range(len(first))]
    if isinstance(first, (int, float, bool)):
        return _collate_pinned(ring, slot, [torch.tensor(v) for v in samples], counter)
    from torch.utils.data._utils.collate import default_collate
    return default_collate(samples)


def _collate_device(bufs, samples, counter, device):
    """default_collate's structure for samples that already live on the device, every tensor stacked into a PERSISTENT
    buffer of this loader (keyed by position, shape, dtype; marked `_gnx_stable`): no allocation per batch, and a step graph
    (graphs.py) can take the buffer as its static input instead of copying the batch a second time.  A batch is therefore
    only valid until the next one is drawn - which is how the training loops use it."""
    first = samples[0]
    if torch.is_tensor(first) and first.is_cuda:
        i = counter[0]
        counter[0] += 1
        key = (i, (len(samples),) + tuple(first.shape), first.dtype)
        buf = bufs.get(key)
        if buf is None:
            buf = bufs[key] = torch.empty(key[1], dtype=first.dtype, device=device)
            buf._gnx_stable = True
        return torch.stack(samples, 0, out=buf)
    if isinstance(first, (list, tuple)):
        return [_collate_device(bufs, [smp[k] for smp in samples], counter, device) for k in range(len(first))]
    from torch.utils.data._utils.collate import default_collate
    counter[0] += 1
    return default_collate(samples)


class DevicePrefetcher:
    """Iterate `loader` one batch ahead of the consumer, delivering device tensors.

        for inputs, labels in DevicePrefetcher(loader, device):
            ...

    depth: batches in flight ahead of the consumer (2 = double buffering).

    Pinned footprint: depth + 1 page-locked copies of one batch per loader (a uint8 128-px Visium array: 3 x 285 MB; float
    patches: 3 x 1 GB), allocated before the first batch is delivered and kept across epochs so that no step waits for
    page-locking.  `close()` releases them (the training loops call it when they return)."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.bytes_moved = 0
        self._ring = None                  # the pinned ring outlives an epoch (one iterator at a time uses it)
        self._ring_busy = False
        self._dev_bufs = {}                # device-resident data: persistent collate buffers
        self._dev_busy = False

    def __len__(self):
        return len(self.loader)

    def close(self):
        """Release the pinned staging ring and the persistent device collate buffers (no iterator may be active)."""
        if not self._ring_busy:
            self._ring = None
        if not self._dev_busy:
            self._dev_bufs = {}

    def _resident(self):
        """True when the loader's data already live on the device.  Decided once per loader from the tensors a
        TensorDataset-like dataset HOLDS (`.tensors`, also through `.datasets` of the stacking datasets): no sample is drawn,
        so nothing is decoded (a PatchGridDataset item is a whole array of JPEGs), no transform runs and no random number
        is consumed.  Any other dataset takes the general path, which passes device-resident batches through untouched."""
        cached = getattr(self.loader, '_gnx_resident', None)
        if cached is None:
            def held(ds, depth=0):
                ts = getattr(ds, 'tensors', None)
                if isinstance(ts, (list, tuple)) and ts and all(torch.is_tensor(t) for t in ts):
                    return list(ts)
                subs = getattr(ds, 'datasets', None)
                if subs is None:
                    subs = [getattr(ds, a) for a in ('image_dataset', 'count_dataset', 'dataset') if hasattr(ds, a)]
                if isinstance(subs, (list, tuple)) and subs and depth < 4:
                    out = []
                    for sub in subs:
                        got = held(sub, depth + 1)
                        if got is None:
                            return None
                        out += got
                    return out
                return None
            flat = held(getattr(self.loader, 'dataset', None))
            cached = bool(flat) and all(t.is_cuda for t in flat)
            try:
                self.loader._gnx_resident = cached
            except Exception:
                pass
        return cached

    def __iter__(self):
        if self.device.type != 'cuda':
            yield from self.loader
            return
        if self._resident():
            if self._dev_busy or not _default_collate_loader(self.loader):
                yield from self.loader
                return
            self._dev_busy = True
            try:
                dataset = self.loader.dataset
                batches = _epoch_index_batches(self.loader)
                from torch.utils.data import TensorDataset
                if type(dataset) is TensorDataset and batches:
                    # A plain TensorDataset on the device: a batch is ONE gather per tensor (index_select into the loader's
                    # persistent buffer) instead of `batch_size` sample views stacked - the same values bit for bit;
                    # default_collate's structure (a list, one stacked tensor per dataset tensor).  The whole epoch's indices
                    # go to the device in one copy.  (A count-MLP spot step at batch 128 is ~0.15 ms of replayed kernels: 128
                    # `dataset[k]` calls + two 128-way stacks were 0.4 ms of host time per batch, the bound of config 1.)
                    flat = torch.tensor([k for b in batches for k in b], dtype=torch.int64).to(self.device)
                    off = 0
                    for b in batches:
                        idx = flat[off:off + len(b)]
                        off += len(b)
                        out = []
                        for i, t in enumerate(dataset.tensors):
                            key = (i, (len(b),) + tuple(t.shape[1:]), t.dtype)
                            buf = self._dev_bufs.get(key)
                            if buf is None:
                                buf = self._dev_bufs[key] = torch.empty(key[1], dtype=t.dtype, device=self.device)
                                buf._gnx_stable = True
                            torch.index_select(t, 0, idx, out=buf)
                            out.append(buf)
                        yield out
                else:
                    for idx in batches:
                        yield _collate_device(self._dev_bufs, [dataset[k] for k in idx], [0], self.device)
            finally:
                self._dev_busy = False
            return
        own_collate = _default_collate_loader(self.loader)
        epoch_batches = iter(_epoch_index_batches(self.loader)) if own_collate else None
        q = queue.Queue(maxsize=self.depth)
        if self._ring_busy:                                       # a second iterator alongside the first: its own ring
            ring, owns = _PinnedRing(self.depth + 1), False
        else:
            if self._ring is None:
                self._ring = _PinnedRing(self.depth + 1)
            ring, owns = self._ring, True
            self._ring_busy = True
            ring.events = [None] * len(ring.slots)
        side = torch.cuda.Stream(device=self.device)
        failure = []
        stop = threading.Event()

        def produce():
            try:
                torch.cuda.set_device(self.device)
                if own_collate:
                    dataset, index_batches = self.loader.dataset, epoch_batches
                else:
                    it = iter(self.loader)
                n = 0
                while not stop.is_set():
                    slot = n % len(ring.slots)
                    if ring.events[slot] is not None:
                        ring.events[slot].synchronize()          # the copies that last read this slot's buffers are done
                    try:
                        if own_collate:                          # samples stacked straight into this slot's pinned buffers
                            batch = _collate_pinned(ring, slot, [dataset[k] for k in next(index_batches)], [0])
                        else:
                            batch = next(it)
                    except StopIteration:
                        break
                    n += 1
                    if n == 1:
                        ring.prime(slot)
                    if all(t.is_cuda for t in _tensors(batch, [])):
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
                    if n == 1:
                        ring.prime(slot)                         # buffers staged by `move` (a loader with its own collate)
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
            if owns:
                if thread.is_alive():
                    self._ring = None                            # still written by a stuck producer: never reuse it
                else:
                    for ev in ring.events:                       # H2D copies out of the ring still in flight
                        if ev is not None:
                            ev.synchronize()
                self._ring_busy = False
        if failure:
            raise failure[0]


def wrap(loader, device):
    """The loops' feed: a DevicePrefetcher on a HIP device, the loader itself otherwise (or when it already is one)."""
    if isinstance(loader, DevicePrefetcher) or torch.device(device).type != 'cuda':
        return loader
    cached = getattr(loader, '_gnx_prefetcher', None)          # one per loader: its pinned ring then serves every epoch
    if isinstance(cached, DevicePrefetcher) and cached.device == torch.device(device):
        return cached
    pf = DevicePrefetcher(loader, device)
    try:
        loader._gnx_prefetcher = pf
    except Exception:                                          # a loader without a __dict__: a fresh wrapper per epoch
        pass
    return pf


def release(dataloaders):
    """Drop the pinned rings / device collate buffers `wrap` attached to these loaders (a dict or an iterable of loaders)."""
    loaders = dataloaders.values() if isinstance(dataloaders, dict) else dataloaders
    for loader in loaders:
        pf = getattr(loader, '_gnx_prefetcher', None)
        if isinstance(pf, DevicePrefetcher):
            pf.close()
