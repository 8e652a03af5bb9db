"""torch.autograd bridges from tensors to the C-ABI kernels (gridnext_amd/csrc, include/gridnext_hip.h).

torch is plumbing here: it owns device memory (caching allocator), streams and the autograd tape;
every arithmetic step below is a hand-written gfx950 kernel.  All matrices are channels-last:
a spot (grid position) is a row, its features are contiguous.
"""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

F32 = torch.float32


def _rows(t):
    """(pointer-compatible 2-D view, leading dimension) of a [M, C] tensor whose rows are contiguous."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t, ld


# ----------------------------------------------------------------------------- hexagonal convolution
class _HexConv(Function):
    @staticmethod
    def forward(ctx, x, kernel0, kernel1, bias, mode):
        x = x.contiguous()
        B, H, W, I = x.shape
        O = kernel0.shape[0]
        k0, k1 = kernel0.contiguous(), kernel1.contiguous()
        y = torch.empty((B, H, W, O), device=x.device, dtype=F32)
        L.call('gnx_hexconv_fwd', L.ptr(x), L.ptr(k0), L.ptr(k1), L.ptr(bias), L.ptr(y), B, H, W, I, O, mode,
               L.stream())
        ctx.save_for_backward(x, k0, k1)
        ctx.mode, ctx.has_bias = mode, bias is not None
        ctx.leaves = (kernel0, kernel1, bias)                  # (the parameters themselves: see _hex_flush)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, k0, k1 = ctx.saved_tensors
        dy = dy.contiguous()
        B, H, W, I = x.shape
        O = k0.shape[0]
        dx = dk0 = dk1 = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.call('gnx_hexconv_bwd_data', L.ptr(dy), L.ptr(k0), L.ptr(k1), L.ptr(dx), B, H, W, I, O, ctx.mode,
                   L.stream())
        if any(ctx.needs_input_grad[1:4]):
            dk0, dk1 = torch.empty_like(k0), torch.empty_like(k1)
            db = torch.empty(O, device=x.device, dtype=F32) if ctx.has_bias else None
            item = (x, dy, dk0, dk1, db, B, H, W, I, O, ctx.mode, ctx.leaves)
            # Inside a step capture (graphs.py: every parameter's .grad is None there, so autograd only STORES what is returned
            # here - as the tensor itself or as a copy of it, no arithmetic) the weight gradients of the backward's hex layers
            # are filled by ONE batched launch at the end of the backward instead of a launch + reduce per layer.  Only for
            # leaf parameters met once in the backward (a weight used twice would have its two tensors ADDED right away).
            leaves = [t for t in ctx.leaves if t is not None]
            if torch.cuda.is_current_stream_capturing() and I <= 32 and O <= 32 and len(_HEX_DEFERRED) < 8 and \
                    all(t.is_leaf and t.grad is None for t in leaves) and \
                    all(t is not u for it in _HEX_DEFERRED for u in it[-1] for t in leaves):
                torch.autograd.Variable._execution_engine.queue_callback(_hex_flush)   # (the first one flushes, the rest find nothing)
                # (the list keeps NO reference to the returned tensors: autograd then keeps them as .grad instead of copying them)
                _HEX_DEFERRED.append((x, dy, None, None, None, B, H, W, I, O, ctx.mode, ctx.leaves))
            else:
                _hex_wgrad_now(item)
        return dx, dk0, dk1, db, None


class _HexWgradItem(ctypes.Structure):
    """gnx_hexconv_wgrad_item of include/gridnext_hip.h."""
    _fields_ = [('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dkernel0', ctypes.c_void_p), ('dkernel1', ctypes.c_void_p),
                ('dbias', ctypes.c_void_p), ('workspace', ctypes.c_void_p), ('B', ctypes.c_int), ('H', ctypes.c_int),
                ('W', ctypes.c_int), ('I', ctypes.c_int), ('O', ctypes.c_int), ('mode', ctypes.c_int),
                ('accumulate', ctypes.c_int), ('pad', ctypes.c_int)]


_HEX_DEFERRED = []          # hex weight gradients of the running (captured) backward, filled by _hex_flush at its end


def _hex_wgrad_now(item):
    x, dy, dk0, dk1, db, B, H, W, I, O, mode, _leaves = item
    ws = torch.empty(L.query('gnx_hexconv_bwd_weight_workspace', B, H, W, I, O), device=x.device, dtype=F32)
    L.call('gnx_hexconv_bwd_weight', L.ptr(x), L.ptr(dy), L.ptr(dk0), L.ptr(dk1), L.ptr(db), L.ptr(ws), B, H, W, I, O, mode, 0,
           L.stream())


def _hex_flush():
    """End of a backward pass: the deferred hex weight gradients as one launch + one batched reduce
    (gnx_hexconv_bwd_weight_batch: bit-identical to the single calls); singly where that entry point declines."""
    items = list(_HEX_DEFERRED)
    _HEX_DEFERRED.clear()
    if not items:
        return
    # The destinations are the parameters' .grad tensors as autograd left them: it either kept the (still unfilled) tensor the
    # backward returned or stored a COPY of it (AccumulateGrad clones a gradient someone else still references - this list
    # does), so the tensors in the items may no longer be the ones anybody reads.
    fixed = []
    for (x, dy, dk0, dk1, db, B, H, W, I, O, mode, leaves) in items:
        dst = []
        for leaf in leaves:
            g = None if leaf is None else leaf.grad
            if leaf is not None and (g is None or not g.is_contiguous() or g.shape != leaf.shape or g.dtype != F32):
                raise RuntimeError("gridnext_amd: a deferred hex-conv weight gradient has no destination (.grad missing or of "
                                   "another layout after the backward pass)")
            dst.append(g)
        fixed.append((x, dy, dst[0], dst[1], dst[2], B, H, W, I, O, mode, leaves))
    items = fixed
    if len(items) > 1:
        arr = (_HexWgradItem * len(items))()
        keep = []
        for a, (x, dy, dk0, dk1, db, B, H, W, I, O, mode, _leaves) in zip(arr, items):
            ws = torch.empty(L.query('gnx_hexconv_bwd_weight_workspace', B, H, W, I, O), device=x.device, dtype=F32)
            keep.append(ws)
            a.x, a.dy, a.dkernel0, a.dkernel1, a.dbias, a.workspace = L.ptr(x), L.ptr(dy), L.ptr(dk0), L.ptr(dk1), L.ptr(db), L.ptr(ws)
            a.B, a.H, a.W, a.I, a.O, a.mode, a.accumulate, a.pad = B, H, W, I, O, mode, 0, 0
        rc = L.query('gnx_hexconv_bwd_weight_batch', ctypes.addressof(arr), len(items), L.stream())
        if rc == 0:
            return
        if rc != L.ERR_UNSUPPORTED:
            raise RuntimeError("gnx_hexconv_bwd_weight_batch failed (%d)" % rc)
    for item in items:
        _hex_wgrad_now(item)


def hexconv(x_nhwc, kernel0, kernel1, bias, oddr):
    """7-neighbour hex conv on a channels-last grid [B, H, W, C]; oddr=True applies it on the Visium
    odd-right grid directly (what gridnet_models.py:178-185 does with rot90/flip copies)."""
    return _HexConv.apply(x_nhwc, kernel0, kernel1, bias, 1 if oddr else 0)


# ----------------------------------------------------------------------------- batch norm (+ReLU)
def bump_versions(*tensors):
    """Tell torch that tensors written through raw pointers changed (host-side only, no launch): whatever is cached on their
    `_version` - the composed affine stages of a frozen MLP below - is then rebuilt."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def bump_batchnorm_versions(modules):
    """The same for every BatchNorm buffer under `modules`: called by the hipGraph steppers after replaying a TRAIN step, whose
    BatchNorm kernels updated running statistics without any Python running."""
    for mod in modules:
        for m in mod.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.running_mean is not None and m.training:
                bump_versions(m.running_mean, m.running_var, m.num_batches_tracked)


class _BNReLU(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps,
                relu, sync=None):
        x, ld = _rows(x)
        M, C = x.shape
        dev = x.device
        stats = torch.empty((4, C), device=dev, dtype=F32)          # scale, shift, mean, invstd
        if training:
            if momentum is None:
                raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not on the GridNext path")
            # statistics and the apply pass in one call (one launch for a matrix of one Visium grid)
            ws = torch.empty(L.query('gnx_bn_workspace', M, C), device=dev, dtype=F32)
            y = torch.empty((M, C), device=dev, dtype=F32)
            L.call('gnx_bn_train_stats_apply_sync', L.ptr(x), ld, M, C, L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                   L.ptr(running_var), L.ptr(num_batches_tracked, torch.int64), float(momentum), float(eps),
                   L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(y), C, 1 if relu else 0,
                   L.ptr(ws), None if sync is None else sync.data_ptr(), L.stream())
            bump_versions(running_mean, running_var, num_batches_tracked)   # (the kernel wrote them through raw pointers)
        else:
            L.call('gnx_bn_fold_eval', C, L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var),
                   float(eps), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), L.stream())
            y = torch.empty((M, C), device=dev, dtype=F32)
            L.call('gnx_scale_shift_relu', L.ptr(x), ld, L.ptr(y), C, M, C, L.ptr(stats[0]), L.ptr(stats[1]),
                   1 if relu else 0, L.stream())
        ctx.save_for_backward(x, stats)
        ctx.cfg = (ld, bool(training), bool(relu), gamma is not None)
        ctx.sync = sync
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        ld, training, relu, affine = ctx.cfg
        M, C = x.shape
        dy, lddy = _rows(dy)
        dev = x.device
        dx = torch.empty((M, C), device=dev, dtype=F32) if ctx.needs_input_grad[0] else None
        want_affine = affine and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dgamma = torch.empty(C, device=dev, dtype=F32) if want_affine else None
        dbeta = torch.empty(C, device=dev, dtype=F32) if want_affine else None
        ws = torch.empty(L.query('gnx_bn_workspace', M, C), device=dev, dtype=F32)
        L.call('gnx_bn_relu_bwd_sync', L.ptr(dy), lddy, L.ptr(x), ld, L.ptr(dx), C, M, C, L.ptr(stats[0]),
               L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(dgamma), L.ptr(dbeta), 1 if relu else 0,
               1 if training else 0, 0, 0, L.ptr(ws), None if ctx.sync is None else ctx.sync.data_ptr(), L.stream())
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def batch_norm_relu(x2d, bn, relu):
    """BatchNorm1d/2d semantics of torch over the rows of x2d [M, C], optionally fused with ReLU.
    Uses (and, in training mode, updates) the module's own parameters and buffers."""
    training = bn.training or not bn.track_running_stats
    if training and getattr(bn, '_gnx_is_sync', False):
        from . import distributed as gdist
        if gdist.sync_active():                     # exact batch-of-`world` emulation: statistics over every rank's rows
            x2 = x2d if x2d.dim() == 2 else x2d.reshape(-1, x2d.shape[-1])
            return gdist.sync_batch_norm_rows(x2, bn, relu)
    return _BNReLU.apply(x2d, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                         training, bn.momentum, bn.eps, relu, _bn_sync(bn, x2d.device))


def _bn_sync(bn, device):
    """The layer's persistent sync words for the several-workgroups-per-channel-block BatchNorm kernels (include/gridnext_hip.h:
    gnx_bn_train_stats_apply_sync): zeroed once, left zero by every launch, owned by this module - two launches that could run at
    the same time never share them.  Made outside graph capture only (a capture that meets a layer without them gets None: the
    kernels then zero a scratch area with a memset node)."""
    hit = bn.__dict__.get('_gnx_sync')
    if hit is not None and hit.device == device:
        return hit
    if device.type != 'cuda' or torch.cuda.is_current_stream_capturing():
        return None
    with torch.no_grad():
        hit = torch.zeros(int(L.query('gnx_bn_sync_words', bn.num_features)), device=device, dtype=torch.int32)
    bn.__dict__['_gnx_sync'] = hit
    return hit


_UNIT = {}


def _unit_vectors(C, device):
    """(ones[C], zeros[C]) on `device`, made once (constants: nothing writes them)."""
    key = (C, str(device))
    hit = _UNIT.get(key)
    if hit is None:
        with torch.no_grad():
            hit = (torch.ones(C, device=device, dtype=F32), torch.zeros(C, device=device, dtype=F32))
        if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            _UNIT[key] = hit                                        # (never cache what a graph capture allocated)
    return hit


class _ReLURows(Function):
    @staticmethod
    def forward(ctx, x):
        x, ld = _rows(x)
        M, C = x.shape
        one, zero = _unit_vectors(C, x.device)                          # (cached: two fill launches per call otherwise)
        y = torch.empty((M, C), device=x.device, dtype=F32)
        L.call('gnx_scale_shift_relu', L.ptr(x), ld, L.ptr(y), C, M, C, L.ptr(one), L.ptr(zero), 1, L.stream())
        ctx.save_for_backward(x, one, zero)
        ctx.ld = ld
        return y

    @staticmethod
    def backward(ctx, dy):
        x, one, zero = ctx.saved_tensors
        M, C = x.shape
        dy, lddy = _rows(dy)
        dx = torch.empty((M, C), device=x.device, dtype=F32)
        ws = torch.empty(L.query('gnx_bn_workspace', M, C), device=x.device, dtype=F32)
        L.call('gnx_bn_relu_bwd', L.ptr(dy), lddy, L.ptr(x), ctx.ld, L.ptr(dx), C, M, C, L.ptr(one), L.ptr(zero),
               L.ptr(zero), L.ptr(one), None, None, 1, 0, 0, 0, L.ptr(ws), L.stream())
        return dx


def relu_rows(x2d):
    return _ReLURows.apply(x2d)


# ----------------------------------------------------------------------------- Linear (fp32 MFMA GEMM)
class _Linear(Function):
    """y[M,N] = A x W^T + b.  `x` is either [M, K] with contiguous rows, or - kmajor - a tensor
    [B, K, S] (spot index contiguous: a (B, genes, H*W) count grid read in place), M = B*S."""

    @staticmethod
    def forward(ctx, x, weight, bias, kmajor):
        w = weight.contiguous()
        N, K = w.shape
        dev = x.device
        if kmajor:
            x = x.contiguous()
            Bn, Kx, S = x.shape
            assert Kx == K
            M = Bn * S
            y = torch.empty((M, N), device=dev, dtype=F32)
            nws = L.query('gnx_gemm_f32_workspace', S, N, K)
            ws = torch.empty(nws, device=dev, dtype=F32) if nws else None
            for b in range(Bn):
                L.call('gnx_gemm_f32_ws', L.ptr(x[b]), S, 1, L.ptr(w), K, 0, L.ptr(bias), L.ptr(y[b * S:]), N,
                       S, N, K, 0, L.ptr(ws), L.stream())
            ctx.ld = S
        else:
            x, ld = _rows(x)
            M = x.shape[0]
            assert x.shape[1] == K
            y = torch.empty((M, N), device=dev, dtype=F32)
            nws = L.query('gnx_gemm_f32_workspace', M, N, K)
            ws = torch.empty(nws, device=dev, dtype=F32) if nws else None
            L.call('gnx_gemm_f32_ws', L.ptr(x), ld, 0, L.ptr(w), K, 0, L.ptr(bias), L.ptr(y), N, M, N, K, 0,
                   L.ptr(ws), L.stream())
            ctx.ld = ld
        ctx.save_for_backward(x, w)
        ctx.kmajor, ctx.has_bias = bool(kmajor), bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, K = w.shape
        dy, lddy = _rows(dy)
        M = dy.shape[0]
        dev = dy.device
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX[M,K] = dY[M,N] . W[N,K]   (B operand stored [k'=n][n'=k] -> b_kmajor)
            dxr = torch.empty((M, K), device=dev, dtype=F32)
            L.call('gnx_gemm_f32', L.ptr(dy), lddy, 0, L.ptr(w), K, 1, None, L.ptr(dxr), K, M, K, N, 0, L.stream())
            if ctx.kmajor:
                Bn, _, S = x.shape
                dx = dxr.view(Bn, S, K).transpose(1, 2)
            else:
                dx = dxr
        if ctx.needs_input_grad[1]:
            # dW[N,K] = dY^T[N,M] . X[M,K]   (A = dY read K-major)
            dw = torch.empty((N, K), device=dev, dtype=F32)
            if ctx.kmajor:
                Bn, _, S = x.shape
                for b in range(Bn):
                    L.call('gnx_gemm_f32', L.ptr(dy[b * S:]), lddy, 1, L.ptr(x[b]), S, 0, None, L.ptr(dw), K,
                           N, K, S, 1 if b else 0, L.stream())
            elif M >= 2048:
                # many spots, few features: the split-M slab kernel of the conv weight gradients (deterministic
                # two-stage sum) instead of a GEMM whose whole grid is N/64 x K/64 <= 16 workgroups
                ws = torch.empty(L.query('gnx_wgrad_workspace', M, N, K, 1), device=dev, dtype=F32)
                L.call('gnx_wgrad_bnrelu', L.ptr(dy), lddy, L.ptr(x), ctx.ld, None, None, L.ptr(dw), L.ptr(ws),
                       M, N, K, 0, 1, 0, 0, L.stream())
            else:
                L.call('gnx_gemm_f32', L.ptr(dy), lddy, 1, L.ptr(x), ctx.ld, 1, None, L.ptr(dw), K, N, K, M, 0,
                       L.stream())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(N, device=dev, dtype=F32)
            ws = torch.empty(L.query('gnx_bn_workspace', M, N), device=dev, dtype=F32)
            L.call('gnx_colsum', L.ptr(dy), lddy, M, N, L.ptr(db), 0, L.ptr(ws), L.stream())
        return dx, dw, db, None


def linear(x, weight, bias, kmajor=False):
    return _Linear.apply(x, weight, bias, kmajor)


# ----------------------------------------------------------------------------- count-MLP as a fused pipeline
_MLP_LAYERS = (nn.Linear, nn.BatchNorm1d, nn.ReLU)


def is_hip_sequential(module):
    """True for an nn.Sequential built only from Linear / BatchNorm1d / ReLU that starts with a Linear -
    the spot head of the tutorials (Tutorial_visium_count.ipynb cell 12)."""
    return (isinstance(module, nn.Sequential) and len(module) > 0 and isinstance(module[0], nn.Linear)
            and all(type(m) in _MLP_LAYERS for m in module))


def _frozen_affine_plan(seq):
    """[(W, b, relu)] - the network as it EVALUATES when nothing in it trains and its BatchNorms run on running statistics
    (the count f of `train_gridwise`'s tutorial recipe: every parameter frozen, `patch_classifier.eval()`, training.py:126): the tutorial MLP
    has no activation between its paired Linears (Tutorial_visium_count.ipynb cell 12), so Linear -> Linear -> BatchNorm1d is
    ONE affine map, W = diag(s) W2 W1, b = s (W2 b1 + b2) + t - 2000 -> 100 instead of 2000 -> 500 -> 100: a fifth of the
    multiply-adds of the layer that is 93 % of f, and three launches instead of nine.  Composed in fp64, rounded once; the
    result differs from the layer-by-layer evaluation by fp32 round-off only (other association order).  Cached on every
    tensor's version; None when the module is not of that shape (a BatchNorm that does not follow a Linear chain)."""
    tensors = list(seq.parameters()) + list(seq.buffers())
    key = tuple((t._version, t.data_ptr()) for t in tensors)
    hit = seq.__dict__.get('_gnx_affine_plan')
    if hit is not None and hit[0] == key:
        return hit[1]
    layers, plan, i = list(seq), [], 0
    with torch.no_grad():
        while i < len(layers):
            m = layers[i]
            if isinstance(m, nn.Linear):
                W = m.weight.double()
                b = m.bias.double() if m.bias is not None else torch.zeros(W.shape[0], device=W.device, dtype=torch.float64)
                i += 1
                while i < len(layers) and isinstance(layers[i], (nn.Linear, nn.BatchNorm1d)):
                    n = layers[i]
                    if isinstance(n, nn.Linear):
                        W2 = n.weight.double()
                        W, b = W2 @ W, W2 @ b + (n.bias.double() if n.bias is not None else 0.0)
                    else:
                        if not n.track_running_stats or n.running_mean is None:
                            plan = None
                            break
                        sc = (n.running_var.double() + n.eps).rsqrt()
                        if n.weight is not None:
                            sc = sc * n.weight.double()
                        sh = (n.bias.double() if n.bias is not None else 0.0) - n.running_mean.double() * sc
                        W, b = sc[:, None] * W, sc * b + sh
                    i += 1
                if plan is None:
                    break
                relu = i < len(layers) and isinstance(layers[i], nn.ReLU)
                i += 1 if relu else 0
                plan.append((W.float().contiguous(), b.float().contiguous(), relu))
            elif isinstance(m, nn.ReLU):
                plan.append((None, None, True))
                i += 1
            else:
                plan = None
                break
    seq.__dict__['_gnx_affine_plan'] = (key, plan)
    return plan


def sequential_forward(seq, x, kmajor=False):
    """Run the user's own nn.Sequential (its parameters, its BN buffers) through the HIP kernels.
    x: [M, K] rows, or with kmajor a [B, K, S] count grid (see _Linear)."""
    # (a network with a trainable parameter is never composed, also not in its no-grad validation passes: the eager and the
    #  graph-replayed loops must evaluate it the same way, bit for bit)
    frozen = not seq.training and not x.requires_grad and not any(p.requires_grad for p in seq.parameters())
    if frozen and getattr(seq, 'fold_frozen', True) and x.is_cuda:
        capturing = torch.cuda.is_current_stream_capturing()
        # Under hipGraph capture the composed weights become constants of the graph: only a network that CANNOT change between
        # replays (no parameter takes gradients) may use them, and only a plan that already exists (made by the eager warm-up
        # batches: nothing composed inside a capture may end up in the cache)
        if capturing and (any(p.requires_grad for p in seq.parameters()) or any(m.training for m in seq.modules()) or
                          seq.__dict__.get('_gnx_affine_plan', (None,))[0] !=
                          tuple((t._version, t.data_ptr()) for t in list(seq.parameters()) + list(seq.buffers()))):
            plan = None
        else:
            plan = _frozen_affine_plan(seq)
        if plan is not None and capturing:
            # the graph keeps raw pointers to these tensors: they must outlive the cache entry they came from
            seq.__dict__.setdefault('_gnx_pinned_plans', []).append(plan)
        if plan is not None:
            first = True
            for W, b, relu in plan:
                if W is not None:
                    x = linear(x, W, b, kmajor if first else False)
                    first = False
                if relu:
                    x = relu_rows(x)
            return x
    layers = list(seq)
    i = 0
    first = True
    while i < len(layers):
        m = layers[i]
        if isinstance(m, nn.Linear):
            x = linear(x, m.weight, m.bias, kmajor if first else False)
            first = False
            i += 1
        elif isinstance(m, nn.BatchNorm1d):
            fuse = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
            x = batch_norm_relu(x, m, relu=fuse)
            i += 2 if fuse else 1
        elif isinstance(m, nn.ReLU):
            x = relu_rows(x)
            i += 1
        else:
            raise TypeError("unsupported layer %r" % type(m))
    return x


# ----------------------------------------------------------------------------- masked cross-entropy
class _MaskedCE(Function):
    @staticmethod
    def forward(ctx, logits, labels, label_base, accum_iters):
        z, ld = _rows(logits)
        M, C = z.shape
        lab = labels.reshape(-1).contiguous()
        if lab.dtype != torch.int64:
            lab = lab.long()
        dev = z.device
        loss = torch.empty((), device=dev, dtype=F32)
        stats = torch.empty(2, device=dev, dtype=torch.int64)
        preds = torch.empty(M, device=dev, dtype=torch.int64)
        ws = torch.empty(L.query('gnx_masked_ce_workspace', M), device=dev, dtype=torch.float64)
        L.call('gnx_masked_ce_fwd', L.ptr(z), ld, L.ptr(lab, torch.int64), M, C, label_base, float(accum_iters),
               L.ptr(loss), L.ptr(stats, torch.int64), L.ptr(preds, torch.int64), L.ptr(ws, torch.float64),
               L.stream())
        ctx.save_for_backward(z, lab, stats)
        ctx.cfg = (ld, label_base, float(accum_iters))
        ctx.mark_non_differentiable(stats, preds)
        return loss, stats, preds

    @staticmethod
    def backward(ctx, dloss, _ds, _dp):
        z, lab, stats = ctx.saved_tensors
        ld, label_base, accum = ctx.cfg
        M, C = z.shape
        dz = torch.empty((M, C), device=z.device, dtype=F32)
        dloss = dloss.contiguous()
        L.call('gnx_masked_ce_bwd', L.ptr(z), ld, L.ptr(lab, torch.int64), M, C, label_base,
               L.ptr(stats, torch.int64), L.ptr(dloss), accum, L.ptr(dz), C, L.stream())
        return dz, None, None, None


def masked_cross_entropy(logits_rows, labels, accum_iters=1, label_base=1):
    """(loss, stats=[n_foreground, n_correct], preds) of training.py:152-160 on channels-last rows [M, C].
    label_base=1: 0 is background (grid loop); label_base=0: plain mean CE (train_spotwise)."""
    return _MaskedCE.apply(logits_rows, labels, label_base, accum_iters)
