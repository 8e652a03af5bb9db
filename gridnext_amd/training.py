"""Training loops of the f∘g path: `train_spotwise` (f alone) and `train_gridwise` (f frozen or not, then g).

Drop-in for /root/reference/gridnext/training.py (:11-98, :101-209): same signatures, same printed lines
('{phase} Loss: {:.4f} Acc: {:.4f}'), same return value (model, val_history, train_history), same side-effect
files (`outfile`, and `<outfile stem>.opt` in the grid loop), and the reference's number-changing quirks:
  * spot loop: zero_grad before every batch; epoch loss = sum(loss*batch)/len(dataset);
  * grid loop: `model.patch_classifier.eval()` in BOTH phases (:126); no zero_grad before the first backward;
    loss divided by `accum_iters` (also in the reported loss, :174); step iff batch_ind % accum_iters == 0
    (batch 0 steps alone); optional second optimiser `f_opt`; accuracy over foreground spots only.

What differs underneath (MI355X-first):
  * on a HIP device with a plain `nn.CrossEntropyLoss()` the permute -> mask-gather -> CE -> argmax chain is ONE
    fused kernel pair on channels-last logits (no dynamic shapes, no host sync per batch); epoch statistics are
    accumulated on the device and read back once per phase;
  * batches are fed by `prefetch.DevicePrefetcher`: pinned staging buffers, `non_blocking` host -> device copies on a
    side stream one batch ahead, event-ordered (the reference's blocking `.to(device)` of pageable memory, :47-51,
    :135-139, sits inside every step); uint8 patches are converted inside the stem kernel;
  * when torch.distributed is initialised with world_size > 1 the loop is data-parallel: gradients of the
    optimised parameters are all-reduced (RCCL over xGMI) before every optimizer step, epoch statistics are
    summed over ranks, rank 0 writes checkpoints.  Shard the arrays with `distributed.ShardedSampler`.
Any other criterion, or CPU tensors, takes the generic path that spells the reference's tensor ops out.
"""
import copy
import os
import time

import torch
import torch.nn as nn

from . import distributed as gdist
from . import functional as GF
from . import graphs
from . import prefetch


def _plain_ce(criterion):
    return (type(criterion) is nn.CrossEntropyLoss and criterion.weight is None and criterion.reduction == 'mean'
            and criterion.ignore_index == -100 and getattr(criterion, 'label_smoothing', 0.0) == 0.0)


def _to_device(inputs, device):
    if isinstance(inputs, list):
        return [t.to(device) for t in inputs]      # multi-modal input to GridNetHexMM
    return inputs.to(device)


class _PhaseMeter:
    """Running loss / correct / counted-spot sums for one phase, kept where the numbers are produced."""

    def __init__(self, device):
        self.device = device
        self.loss = 0.0            # python float (host path) or 0-dim float64 tensor (device path)
        self.correct = 0
        self.counted = 0
        self.items = 0             # dataset items this rank actually saw (wrap-around duplicates of a sharded sampler included)
        self.acc = None            # [loss, correct, counted] sums on the device (the HIP loops' path: gnx_meter_add)

    def add(self, loss, batch_size, correct, counted):
        self.items += batch_size
        if (torch.is_tensor(loss) and loss.is_cuda and loss.dtype == torch.float32 and loss.numel() == 1 and
                torch.is_tensor(correct) and correct.is_cuda and correct.dtype == torch.int64 and correct.numel() == 1 and
                (isinstance(counted, int) or (torch.is_tensor(counted) and counted.is_cuda and counted.dtype == torch.int64 and
                                              counted.numel() == 1))):
            # the HIP loops' case: ONE launch on three device-resident double sums (the same arithmetic: the fp32 loss widened,
            # times the batch size, added in double; counts are exact in double) instead of five elementwise launches
            if self.acc is None:
                self.acc = torch.zeros(3, device=loss.device, dtype=torch.float64)
            from . import _lib as L
            L.call('gnx_meter_add', self.acc.data_ptr(), loss.data_ptr(), float(batch_size), correct.data_ptr(),
                   None if isinstance(counted, int) else counted.data_ptr(), float(counted) if isinstance(counted, int) else 0.0,
                   L.stream())
            return
        if torch.is_tensor(loss):
            loss = loss.detach().double()
        self.loss = self.loss + loss * batch_size
        self.correct = self.correct + correct
        self.counted = self.counted + counted

    def totals(self):
        vals = [float(v) for v in (self.loss, self.correct, self.counted, self.items)]   # the only host sync of a phase
        if self.acc is not None:
            a = self.acc.tolist()
            vals = [vals[0] + a[0], vals[1] + a[1], vals[2] + a[2], vals[3]]
        return gdist.allreduce_sums(vals, self.device)

    def n_items(self, dataset_len, items_seen):
        """The divisor of the epoch loss: len(dataset) as in the reference (training.py:73, :178); data-parallel, the number
        of items the ranks actually processed - a ShardedSampler pads the tail so every rank gets the same count, and those
        duplicates are in the loss sum."""
        return int(items_seen) if gdist.is_active() else dataset_len


class _BestKeeper:
    """`best_model_wts = copy.deepcopy(model.state_dict())` of the reference (training.py:29, :87-89, :121, :197-199) - a
    snapshot of every parameter and buffer whenever the validation loss improves.  The snapshot lives in buffers allocated once
    and is refreshed with one multi-tensor copy per (device, dtype) instead of one small copy per entry: a count-only grid model
    holds 47 entries (~8 us of device queue each, a tenth of an epoch of ten arrays), DenseNet-121 more than 700."""

    def __init__(self, model):
        self.model = model
        self.best_loss = float('inf')
        self.best_wts = None
        self._snapshot()

    def _snapshot(self):
        sd = self.model.state_dict()
        tensors = {k: v for k, v in sd.items() if torch.is_tensor(v)}
        best = self.best_wts
        stale = best is None or list(best.keys()) != list(sd.keys()) or any(
            (not torch.is_tensor(best[k])) or best[k].shape != v.shape or best[k].dtype != v.dtype or best[k].device != v.device
            for k, v in tensors.items())
        if stale or len(tensors) != len(sd):
            self.best_wts = copy.deepcopy(sd)                 # first snapshot (or an unusual state_dict): the reference's way
            return
        groups = {}
        for k, v in tensors.items():
            d, srcs = groups.setdefault((v.device, v.dtype), ([], []))
            d.append(best[k])
            srcs.append(v.detach())
        with torch.no_grad():
            for dsts, srcs in groups.values():
                torch._foreach_copy_(dsts, srcs)

    def offer(self, epoch_loss):
        if epoch_loss < self.best_loss:
            self.best_loss = epoch_loss
            self._snapshot()
            return True
        return False


class _F16StepGuard:
    """Skip-step on fp16 gradient overflow (new with the fp16-MFMA gradient path; the fp32 reference needs none).

    A DenseNet whose backward ran on the fp16 kernels (`densenet_train_f16`) owns a sticky device flag, `f16_grad_overflow`,
    that any of those kernels sets when it reduces a non-finite value.  `ok()` is called right before the optimizers step:
    it reads AND clears the flags (one 4-byte read-back per optimizer step, only once a model has used that path; with
    several ranks the flag is MAX-reduced first so that every rank takes the same decision).  On overflow the caller skips
    both optimizers and drops the accumulated gradients; the loss-scale target of the affected network
    (`f16_grad_target`, the exponent the largest gradient element is centred on: 12) goes down by one - half the scale - and
    comes back up by one after `GROWTH_INTERVAL` consecutive clean steps, as torch's GradScaler does."""
    GROWTH_INTERVAL = 200
    TARGET_MAX, TARGET_MIN = 12.0, 2.0

    def __init__(self, model):
        self.model = model
        self.skipped = 0
        self.clean = 0

    def _nets(self):
        # (the module tree is walked once: this runs before every optimizer step, and a count-only grid step is ~250 us)
        cands = self.__dict__.get('_cands')
        if cands is None:
            from .densenet import DenseNet
            cands = self._cands = [m for m in self.model.modules() if isinstance(m, DenseNet)]
        return [m for m in cands if m.__dict__.get('f16_grad_overflow') is not None]

    def ok(self):
        nets = self._nets()
        if not nets:
            return True
        flags = torch.stack([m.__dict__['f16_grad_overflow'].reshape(()) for m in nets])
        if gdist.is_active():
            torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MAX)
        hit = flags.tolist()
        if not any(hit):
            self.clean += 1
            if self.clean >= self.GROWTH_INTERVAL:
                self.clean = 0
                for m in nets:
                    m.f16_grad_target = min(self.TARGET_MAX, float(m.__dict__.get('f16_grad_target', self.TARGET_MAX)) + 1.0)
            return True
        self.clean = 0
        self.skipped += 1
        for m, h in zip(nets, hit):
            m.__dict__['f16_grad_overflow'].zero_()
            if h:
                m.f16_grad_target = max(self.TARGET_MIN, float(m.__dict__.get('f16_grad_target', self.TARGET_MAX)) - 1.0)
        if gdist.rank() == 0:
            print('fp16 gradient overflow: optimizer step skipped, loss-scale target lowered to 2^%d'
                  % int(min(float(m.__dict__.get('f16_grad_target', self.TARGET_MAX)) for m in nets)), flush=True)
        return False


def _banner(epoch, num_epochs):
    if gdist.rank() == 0:
        print('Epoch {}/{}'.format(epoch, num_epochs - 1), flush=True)
        print('-' * 10, flush=True)


def _report(phase, loss, acc):
    if gdist.rank() == 0:
        print('{} Loss: {:.4f} Acc: {:.4f}'.format(phase, loss, acc), flush=True)


def _finish(since, best_loss):
    if gdist.rank() == 0:
        elapsed = time.time() - since
        print('Training complete in {:.0f}m {:.0f}s'.format(elapsed // 60, elapsed % 60), flush=True)
        print('Best val loss: {:4f}'.format(best_loss), flush=True)


# --------------------------------------------------------------------------------------------------- spot loop
def train_spotwise(model, dataloaders, criterion, optimizer, num_epochs=10, outfile=None, display=False):
    since = time.time()
    val_history, train_history = [], []
    keeper = _BestKeeper(model)
    device = gdist.default_device()
    model.to(device)
    fused_ok = _plain_ce(criterion)
    guard = _F16StepGuard(model)
    hip_mlp = GF.is_hip_sequential(model)
    # an MLP step is ~40 kernels of a few microseconds: launched one by one the loop is bound by the host (1 ms per batch of
    # 128 against ~0.15 ms of kernel time).  Forward, fused CE and backward of each (phase, batch shape) are captured into a
    # hipGraph and replayed (graphs.py); zero_grad, all-reduce, optimizer and statistics stay eager
    stepper = None
    if fused_ok and graphs.wanted_spotwise(model, device):
        def _spot_step(inputs, labels):
            outputs = (GF.sequential_forward(model, inputs.reshape(inputs.shape[0], -1)) if hip_mlp else model(inputs))
            loss, stats, _ = GF.masked_cross_entropy(outputs, labels, 1, label_base=0)
            return loss, stats[1], None
        stepper = graphs.GridStepGraphs(_spot_step, model.parameters(),
                                        drop_derived=getattr(model, 'invalidate_cache', None), models=(model,))

    for epoch in range(num_epochs):
        _banner(epoch, num_epochs)
        for phase in ('train', 'val'):
            model.train(phase == 'train')
            meter = _PhaseMeter(device)
            loader = dataloaders[phase]
            if hasattr(getattr(loader, 'sampler', None), 'set_epoch'):
                loader.sampler.set_epoch(epoch)
            loader = prefetch.wrap(loader, device)         # pinned, double-buffered H2D one batch ahead (no-op on the CPU)
            if display:
                from tqdm import tqdm
                loader = tqdm(loader)
            for inputs, labels in loader:
                batch_size = labels.size(0)
                inputs, labels = _to_device(inputs, device), labels.to(device)
                optimizer.zero_grad()
                replayed = None
                # (a DenseNet step keeps its whole tape in the graph's memory pool: only batches that are launch-bound anyway -
                # up to 64 patches of 128 px - are graphed; larger ones are GPU-bound on large kernels and stay eager)
                if stepper is not None and torch.is_tensor(inputs) and inputs.is_cuda and \
                        (hip_mlp or inputs.numel() <= 64 * 3 * 128 * 128):
                    replayed = stepper.run(phase == 'train', inputs, labels)
                if replayed is not None:
                    loss, correct, _ = replayed
                    on_device = True
                else:
                    with torch.set_grad_enabled(phase == 'train'):
                        if hip_mlp and torch.is_tensor(inputs) and inputs.is_cuda:
                            outputs = GF.sequential_forward(model, inputs.reshape(batch_size, -1))
                        else:
                            outputs = model(inputs)
                        if fused_ok and outputs.is_cuda and outputs.dim() == 2:
                            loss, stats, _ = GF.masked_cross_entropy(outputs, labels, 1, label_base=0)
                            correct = stats[1]
                        else:
                            loss = criterion(outputs, labels)
                            correct = torch.sum(torch.max(outputs, 1)[1] == labels.data)
                        if phase == 'train':
                            loss.backward()
                    on_device = outputs.is_cuda
                    del outputs
                if phase == 'train':
                    if guard.ok():
                        gdist.allreduce_gradients(gdist.optimizer_params(optimizer))
                        optimizer.step()
                    else:
                        gdist.discard_step()
                meter.add(loss if on_device else loss.item(), batch_size, correct, batch_size)
                loss = correct = None
            loss_sum, n_right, _, seen = meter.totals()
            n_items = meter.n_items(len(dataloaders[phase].dataset), seen)
            epoch_loss, epoch_acc = loss_sum / n_items, n_right / n_items
            _report(phase, epoch_loss, epoch_acc)
            if phase == 'val':
                if keeper.offer(epoch_loss) and outfile is not None and gdist.rank() == 0:
                    torch.save(model.state_dict(), outfile)
                val_history.append(epoch_loss)
            else:
                train_history.append(epoch_loss)
        if gdist.rank() == 0:
            print()
    _finish(since, keeper.best_loss)
    prefetch.release(dataloaders)                   # the loaders' pinned staging rings
    model.load_state_dict(keeper.best_wts)
    return model, val_history, train_history


# --------------------------------------------------------------------------------------------------- grid loop
def _grid_loss(model, inputs, labels, criterion, accum_iters, fused_ok):
    """(loss, n_correct, n_foreground) for one batch of arrays."""
    if fused_ok and hasattr(model, 'forward_nhwc') and labels.is_cuda:
        logits = model.forward_nhwc(inputs)                         # [B, H, W, C] channels-last
        assert logits.shape[1] == labels.shape[1] and logits.shape[2] == labels.shape[2], \
            "Output tensor does not match label dimensions!"
        loss, stats, _ = GF.masked_cross_entropy(logits.reshape(-1, logits.shape[-1]), labels, accum_iters,
                                                 label_base=1)
        return loss, stats[1], stats[0]
    outputs = model(inputs)
    assert outputs.shape[2] == labels.shape[1] and outputs.shape[3] == labels.shape[2], \
        "Output tensor does not match label dimensions!"
    rows = outputs.permute(0, 2, 3, 1).reshape(-1, outputs.shape[1])
    flat = labels.reshape(-1)
    keep = flat > 0
    rows, flat = rows[keep], flat[keep] - 1                         # foreground classes are 1..N_CLASS
    loss = criterion(rows, flat) / accum_iters
    correct = torch.sum(torch.max(rows, 1)[1] == flat)
    return loss, correct, flat.numel()


def train_gridwise(model, dataloaders, criterion, optimizer, num_epochs=10, outfile=None,
                   f_opt=None, accum_iters=1):
    since = time.time()
    train_history, val_history = [], []
    keeper = _BestKeeper(model)
    device = gdist.default_device()
    model.to(device)
    fused_ok = _plain_ce(criterion)
    stepped = gdist.optimizer_params(optimizer, f_opt)
    guard = _F16StepGuard(model)
    # launch-bound models (count-only f + g: ~60 kernels of 5-20 us per array): the step is captured once per phase and input
    # shape into a hipGraph and replayed (graphs.py); optimizer, all-reduce and statistics stay eager
    stepper = None
    if graphs.wanted(model, fused_ok, device) and not gdist.sync_active():    # (collectives inside the step: not capturable)
        stepper = graphs.GridStepGraphs(lambda i, l: _grid_loss(model, i, l, criterion, accum_iters, fused_ok),
                                        model.parameters(), models=(model,))

    for epoch in range(num_epochs):
        _banner(epoch, num_epochs)
        for phase in ('train', 'val'):
            model.train(phase == 'train')
            model.patch_classifier.eval()           # f's BN/dropout stay frozen in both phases
            meter = _PhaseMeter(device)
            loader = dataloaders[phase]
            if hasattr(getattr(loader, 'sampler', None), 'set_epoch'):
                loader.sampler.set_epoch(epoch)
            loader = prefetch.wrap(loader, device)         # pinned, double-buffered H2D one batch ahead (no-op on the CPU)
            loss = None
            for batch_ind, (inputs, labels) in enumerate(loader):
                loss = correct = n_fg = None        # (drops the previous step's autograd graph before a capture)
                batch_size = labels.size(0)
                inputs, labels = _to_device(inputs, device), labels.to(device)
                with torch.set_grad_enabled(phase == 'train'):
                    replayed = stepper.run(phase == 'train', inputs, labels) if stepper is not None else None
                    if replayed is not None:
                        loss, correct, n_fg = replayed              # forward, loss and (train) backward: one graph launch
                    else:
                        loss, correct, n_fg = _grid_loss(model, inputs, labels, criterion, accum_iters, fused_ok)
                        loss = gdist.global_foreground_mean(loss, n_fg)    # (identity unless the exact-batch mode is on)
                        if phase == 'train':
                            loss.backward()
                    if phase == 'train':
                        if batch_ind % accum_iters == 0:
                            if guard.ok():
                                gdist.allreduce_gradients(stepped)
                                optimizer.step()
                                if f_opt is not None:
                                    f_opt.step()
                            else:                           # a non-finite fp16 gradient: nothing of this step is applied
                                gdist.discard_step()
                            optimizer.zero_grad()
                            if f_opt is not None:
                                f_opt.zero_grad()
                meter.add(loss if labels.is_cuda else loss.item(), batch_size, correct, n_fg)
            loss_sum, n_right, n_fg_total, seen = meter.totals()
            epoch_loss = loss_sum / meter.n_items(len(dataloaders[phase].dataset), seen)
            epoch_acc = n_right / n_fg_total if n_fg_total else float('nan')
            _report(phase, epoch_loss, epoch_acc)
            if phase == 'val':
                if keeper.offer(epoch_loss) and outfile is not None and gdist.rank() == 0:
                    torch.save(model.state_dict(), outfile)
                    opt_file = os.path.splitext(outfile)[0] + ".opt"
                    if f_opt is not None:
                        torch.save({'g_opt': optimizer.state_dict(), 'f_opt': f_opt.state_dict()}, opt_file)
                    else:
                        torch.save(optimizer.state_dict(), opt_file)
                val_history.append(epoch_loss)
            else:
                train_history.append(epoch_loss)
        if gdist.rank() == 0:
            print()
    _finish(since, keeper.best_loss)
    prefetch.release(dataloaders)                   # the loaders' pinned staging rings
    model.load_state_dict(keeper.best_wts)
    return model, val_history, train_history
