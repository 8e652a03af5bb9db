"""hipGraph replay of the grid loop's step (new; SURVEY 7 "HIP streams and graphs instead of a tracing compiler").

A count-only f + g step (BASELINE config 3) is ~60 kernels of 5-20 us: launched one by one from Python the step is bound
by the host (0.86 ms per array against ~0.5 ms of kernel time).  `GridStepGraph` captures, per (phase, input shapes), ONE
step of `train_gridwise` - f over the grid, g, the fused masked CE and, in the train phase, the whole backward - into a
hipGraph (through `torch.cuda.CUDAGraph`: the C-ABI kernels are launched on torch's current stream, which is the capture
stream) and replays it for every later batch of that shape:

  * the first `WARMUP` batches of a shape run eagerly (they size caches, fold BatchNorm constants, repack weights);
  * batch tensors are copied into static input buffers (40 MB of counts: microseconds); outputs are static tensors;
  * the captured backward writes fresh gradients into graph-owned tensors; after each replay they are handed to
    `p.grad` by copy (`p.grad = g.clone()` or `p.grad += g`), so the reference's accumulation semantics (no zero_grad
    before the first backward, `accum_iters`) hold and nothing aliases graph memory;
  * optimizer steps, gradient all-reduce and epoch statistics stay outside the graph (eager, unchanged).

Used only where it is safe AND pays: every parameter of the spot classifier(s) that takes gradients is a plain MLP or
frozen - a DenseNet on the gradient path keeps host-side state per forward (tape checks, cache invalidation) and its
steps are GPU-bound anyway.  `GNX_GRAPH=0` disables it, `GNX_GRAPH=1` forces it for every fused-path model.
"""
import os

import torch

WARMUP = 2          # eager batches per (phase, shapes) before capture
MAX_GRAPHS = 6      # distinct (phase, shapes) captured per loop call; further shapes stay eager


def _shapes(obj):
    if torch.is_tensor(obj):
        return (tuple(obj.shape), obj.dtype)
    if isinstance(obj, (list, tuple)):
        return tuple(_shapes(o) for o in obj)
    return None


def _static_like(obj):
    if torch.is_tensor(obj):
        # a batch stacked into a persistent buffer of its loader (prefetch._collate_device) IS static: the graph reads it
        # in place and the second copy of the batch (40 MB for a count grid) does not happen
        return obj if getattr(obj, '_gnx_stable', False) else torch.empty_like(obj)
    return [_static_like(o) for o in obj]


def _copy_into(dst, src):
    if torch.is_tensor(dst):
        if dst is not src and not (dst.data_ptr() == src.data_ptr() and dst.shape == src.shape):
            dst.copy_(src, non_blocking=True)
    else:
        for d, s in zip(dst, src):
            _copy_into(d, s)


def wanted(model, fused_ok, device):
    """Graph the grid step of this model?"""
    flag = os.environ.get('GNX_GRAPH', '')
    if flag == '0' or not fused_ok or torch.device(device).type != 'cuda' or not hasattr(model, 'forward_nhwc'):
        return False
    from .densenet import DenseNet
    from . import functional as GF
    fs = [getattr(model, n) for n in ('image_classifier', 'count_classifier') if hasattr(model, n)] or \
        [model.patch_classifier]
    for f in fs:
        if isinstance(f, DenseNet):
            if f.training or any(p.requires_grad for p in f.parameters()):
                # tape, version checks, cache invalidation, a host read of the gamma != 0 check: host-side state that a
                # capture cannot hold - refused even under GNX_GRAPH=1 (the grid stepper has no drop_derived hook)
                return False
            if flag != '1':
                return False                                   # frozen DenseNet: the step is GPU-bound, nothing to gain
            continue
        if not GF.is_hip_sequential(f) and flag != '1':
            return False                                       # an arbitrary user module: do not guess
    return True


def wanted_spotwise(model, device):
    """Graph the spot loop's step of this model?  An MLP classifier or a DenseNet on a HIP device (GNX_GRAPH=0 turns it
    off).  A DenseNet step at the tutorial's batch of 32 is ~1 200 launches whose enqueue time equals their GPU time; replayed,
    the host is out of the step.  The stepper is given `model.invalidate_cache` (GridStepGraphs.drop_derived)."""
    if os.environ.get('GNX_GRAPH', '') == '0' or torch.device(device).type != 'cuda':
        return False
    from . import functional as GF
    from .densenet import DenseNet
    if isinstance(model, DenseNet):
        # the recompute path (efficient=True) and train-mode dropout keep host-side state a capture cannot hold: the former
        # saves / restores BatchNorm buffers and the device RNG state around its second forward, which is illegal while
        # capturing - those models step eagerly
        return model.mfma == 'f32' and not model.efficient and not model.drop_rate > 0
    return GF.is_hip_sequential(model)


class GridStepGraph:
    """One captured step for one (phase, shapes).  `step_fn(inputs, labels) -> (loss, correct, n_fg)` is the eager step
    WITHOUT the backward; `train` adds `loss.backward()` to the capture."""

    def __init__(self, step_fn, params, train, models=()):
        self.step_fn, self.train = step_fn, train
        self.models = tuple(models)
        self.params = [p for p in params if p.requires_grad] if train else []
        self.graph = None
        self.out_grads = None
        self.bn_buffers = None
        self.seen = 0
        self.failed = False

    def ready(self):
        return self.graph is not None

    def capture(self, inputs, labels):
        self.s_inputs, self.s_labels = _static_like(inputs), _static_like(labels)
        _copy_into(self.s_inputs, inputs)
        _copy_into(self.s_labels, labels)
        keep = [p.grad for p in self.params]
        for p in self.params:
            p.grad = None                                      # the captured backward then CREATES its gradient tensors
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                with torch.set_grad_enabled(self.train):
                    loss, correct, n_fg = self.step_fn(self.s_inputs, self.s_labels)
                    if self.train:
                        loss.backward()
            self.outs = (loss.detach(), correct, n_fg)
            self.s_grads = [p.grad for p in self.params]       # graph-owned; None where a parameter got no gradient
            self.graph = graph                                 # only a COMPLETED capture makes this entry ready()
        except Exception:
            self.failed = True                                 # this (phase, shapes) stays eager for the rest of the call
            raise
        finally:
            for p, g in zip(self.params, keep):
                p.grad = g                                     # capture ran nothing: the caller's gradients are untouched

    def replay(self, inputs, labels):
        """Returns the step's (loss, correct, n_fg) as the graph's STATIC output tensors: valid until the next replay of
        this graph - the loop folds them into its phase sums right away (stream order keeps that safe)."""
        _copy_into(self.s_inputs, inputs)
        _copy_into(self.s_labels, labels)
        self.graph.replay()
        if self.train:
            self._deliver_gradients()
            if self.models:
                # train-mode BatchNorm kernels ran inside the graph: their buffers changed without any Python running.  The
                # buffers are the ones that were in train mode at CAPTURE time - found once, not by walking the module tree after
                # every replay (35 us of a ~250-us count-only step)
                from .functional import bump_versions
                if self.bn_buffers is None:
                    import torch.nn as nn
                    self.bn_buffers = [t for mod in self.models for m in mod.modules()
                                       if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.running_mean is not None and m.training
                                       for t in (m.running_mean, m.running_var, m.num_batches_tracked)]
                bump_versions(*self.bn_buffers)
        return self.outs

    def _deliver_gradients(self):
        """Graph-owned gradients -> `p.grad`, the reference's accumulation semantics kept (`p.grad` None: becomes this
        step's gradient; otherwise: += it), with ONE multi-tensor kernel per case instead of a copy per parameter (14
        copies of ~5 us were a tenth of a count-only step).  The tensors handed out are the stepper's own and are reused
        by the next replay - like the gradients of an ordinary backward, they are only meaningful until then."""
        if self.out_grads is None:
            self.out_grads = [None if g is None else torch.empty_like(g) for g in self.s_grads]
        set_dst, set_src, acc_dst, acc_src = [], [], [], []
        for p, g, mine in zip(self.params, self.s_grads, self.out_grads):
            if g is None:
                continue
            if p.grad is None:
                set_dst.append(mine)
                set_src.append(g)
                p.grad = mine
            else:                                              # also when p.grad still IS `mine` (no zero_grad in between)
                acc_dst.append(p.grad)
                acc_src.append(g)
        if set_dst:
            torch._foreach_copy_(set_dst, set_src)
        if acc_dst:
            torch._foreach_add_(acc_dst, acc_src)


class GridStepGraphs:
    """The per-loop-call collection: `run(phase_is_train, inputs, labels, eager_fn)`."""

    def __init__(self, step_fn, params, drop_derived=None, models=()):
        """drop_derived: called right before a capture and after every replay - for a model that caches tensors derived from
        its weights (DenseNet: folded BatchNorm vectors, re-laid-out weights).  Before a capture, so that they are recomputed
        INSIDE it from the live parameters (a replay after an optimizer step must not multiply with capture-time copies);
        after a replay, so that nothing outside the graph keeps using tensors that live in its memory pool."""
        self.step_fn, self.params = step_fn, list(params)
        self.drop_derived = drop_derived
        self.models = tuple(models)     # modules whose DenseNets' derived caches are dropped when a capture aborts
        self.table = {}

    def _drop_after_failed_capture(self):
        from . import functional as GF
        GF._HEX_DEFERRED.clear()                               # (weight gradients an aborted backward left unflushed)
        if self.drop_derived is not None:
            self.drop_derived()
        from .densenet import DenseNet
        for obj in self.models:                                # steppers without a drop_derived hook (the grid loop)
            for m in obj.modules():
                if isinstance(m, DenseNet):
                    m.invalidate_cache()

    def run(self, train, inputs, labels):
        """(loss, correct, n_fg, did_backward): graph replay when one exists for this phase and these shapes, else None."""
        key = (bool(train), _shapes(inputs), _shapes(labels))
        st = self.table.get(key)
        if st is None:
            if len(self.table) >= MAX_GRAPHS:
                return None
            st = self.table[key] = GridStepGraph(self.step_fn, self.params, train, self.models)
        if st.failed:
            return None
        if not st.ready():
            st.seen += 1
            if st.seen <= WARMUP:
                return None                                    # eager warm-up batch
            if self.drop_derived is not None:
                self.drop_derived()
            try:
                st.capture(inputs, labels)
            except Exception as exc:                           # e.g. a host synchronisation inside the step: stay eager
                import warnings
                warnings.warn("gridnext_amd.graphs: step capture failed (%s); this step shape runs eagerly" % (exc,))
                # tensors derived inside the aborted capture (folded BatchNorm vectors, re-laid-out weights) were recorded,
                # never executed: the eager step must not find them in the model's cache under still-valid keys
                self._drop_after_failed_capture()
                return None
        out = st.replay(inputs, labels)
        if self.drop_derived is not None:
            self.drop_derived()
        return out
