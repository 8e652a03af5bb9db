"""DenseNet-BC forward-with-tape and backward on the HIP kernels (training / gradient path).

Used by `DenseNet.forward` whenever the module is in training mode or a gradient is required - i.e.
`train_spotwise` on an image classifier (BatchNorm batch statistics, /root/reference/gridnext/training.py:60-67)
and `train_gridwise` with `f_opt` (f kept in eval mode by training.py:126, but its parameters trained).

MI355X-first: a 128-px array's full tape is ~55 GB of the 288 GB HBM, so by default one tape covers the whole batch and
nothing is recomputed.  Where the reference saves memory (`efficient=True`, densenet.py:36-40; `atonce_patch_limit` chunks
under `cp.checkpoint`, gridnet_models.py:88-104) or where the tape would exceed `DenseNet.tape_budget` (256-px arrays), the
batch goes through in chunks whose forward keeps no tape and whose backward rebuilds it (`_RecomputeFn`).
The tape holds the raw (pre-BN) tensors only: stem output, the dense-block buffers and each layer's 1x1-conv output (stored
activated under running statistics); every other BN+ReLU is re-evaluated inside the consuming kernel's operand load, forward
and backward.  One autograd node covers the whole network; gradients are returned for every parameter that requires one.
`DenseNet.mfma = 'f16'` under running statistics takes the fp16 tape and the fp16-MFMA backward of densenet_train_f16.
"""
import ctypes
import os
import struct

import torch
from torch.autograd import Function

from . import _lib as L
from .functional import _bn_sync


class _WgradItem(ctypes.Structure):
    """gnx_wgrad_item of include/gridnext_hip.h: one weight gradient of a batch (gnx_wgrad_bnrelu_batch)."""
    _fields_ = [('dY', ctypes.c_void_p), ('lddy', ctypes.c_long), ('X', ctypes.c_void_p), ('ldx', ctypes.c_long),
                ('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p), ('dW', ctypes.c_void_p), ('workspace', ctypes.c_void_p),
                ('M', ctypes.c_long), ('N', ctypes.c_int), ('K', ctypes.c_int), ('S', ctypes.c_int), ('accumulate', ctypes.c_int)]

F32 = torch.float32


def _cols(t, c0):
    """Device address of column c0 of a row-major [M, ld] matrix."""
    return t.data_ptr() + 4 * c0


class _Tape:
    pass


def _bn(model_bn, X_ptr, ld, M, training, dev, st):
    """Fold one BatchNorm over rows of X[M][C]: returns stats[4][Cpad] = scale, shift, mean, invstd."""
    C = model_bn.num_features
    cp = (C + 3) // 4 * 4
    stats = torch.empty((4, cp), device=dev, dtype=F32)
    if training:
        ws = torch.empty(L.query('gnx_bn_workspace', M, C), device=dev, dtype=F32)
        sync = _bn_sync(model_bn, torch.device(dev))           # (the layer's persistent sync words: no memset node per call)
        L.call('gnx_bn_train_stats_sync', X_ptr, ld, M, C, L.ptr(model_bn.weight), L.ptr(model_bn.bias),
               L.ptr(model_bn.running_mean), L.ptr(model_bn.running_var),
               L.ptr(model_bn.num_batches_tracked, torch.int64), float(model_bn.momentum), float(model_bn.eps),
               L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(ws),
               sync.data_ptr() if sync is not None else None, st)
    else:
        L.call('gnx_bn_fold_eval', C, L.ptr(model_bn.weight), L.ptr(model_bn.bias), L.ptr(model_bn.running_mean),
               L.ptr(model_bn.running_var), float(model_bn.eps), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]),
               L.ptr(stats[3]), st)
    return stats


def relayout_weights(model, kind, dev, st):
    """{weight: re-laid-out copy} for every conv weight of one kind (0: conv2 -> [tap][N][K], 1: conv2 -> [8 - tap][K][N] for
    the data gradient, 2: conv1 / transition conv -> [K][N]) in ONE launch (`gnx_relayout_weights_batch`; a training step
    needs all three after every optimizer step - 3 x 58 small launches otherwise).  Destinations and the device-side pointer
    table persist with the model (rebuilt when a weight's storage moves); the CONTENT is refreshed by every call."""
    layers = [l for _, ls, _, _ in model._blocks for l in ls]
    if kind == 2:
        weights = [l.conv1.weight for l in layers] + [t.conv.weight for _, _, t, _ in model._blocks if t is not None]
    else:
        weights = [l.conv2.weight for l in layers]
    key = (str(dev),) + tuple(w.data_ptr() for w in weights)
    plans = model.__dict__.setdefault('_relayout_plans', {})
    hit = plans.get(kind)
    if hit is None or hit[0] != key:
        dsts, rows = {}, []
        for w in weights:
            # the device table holds RAW pointers read as dense fp32 [N][K][taps]: anything else (channels_last weights
            # after model.to(memory_format=...), half weights) would be re-laid-out silently wrong
            if not (w.is_contiguous() and w.dtype == F32):
                raise TypeError("gridnext_amd.DenseNet: conv weights must be contiguous float32 (got %s, strides %s)"
                                % (w.dtype, tuple(w.stride())))
            n, k = w.shape[0], w.shape[1]
            shape = (9, n, k) if kind == 0 else ((9, k, n) if kind == 1 else (k, n))
            d = torch.empty(shape, device=dev, dtype=F32)
            dsts[w] = d
            rows.append(struct.pack('<QQii', w.data_ptr(), d.data_ptr(), n, k))
        table = torch.frombuffer(bytearray(b''.join(rows)), dtype=torch.uint8).to(dev)
        hit = plans[kind] = (key, dsts, table, len(weights))
    L.call('gnx_relayout_weights_batch', hit[2].data_ptr(), hit[3], kind, st)
    return hit[1]


def gammas_nonzero(model):
    """True when no norm2 (and norm0) weight of the network is exactly zero (the activated-bottleneck form of the training
    forward cannot recover x_hat where gamma == 0).  One device reduction and one host read per change of those weights: the
    answer is cached with the model's other derived tensors, keyed on the weights' versions."""
    ws = [l.norm2.weight for _, ls, _, _ in model._blocks for l in ls]
    if hasattr(model.features, 'norm0'):
        ws.append(model.features.norm0.weight)       # pool0's adjoint reads norm0's mask and x_hat off the pooled map
    key = model._key(ws)
    hit = model._cache.get('g2nz')
    if hit is not None and hit[0] == key:
        return hit[1]
    w = torch.cat([t.detach().reshape(-1) for t in ws])
    ok = bool((w != 0).all().item())
    model._cache['g2nz'] = (key, ok)
    return ok


def _dropout_keep(model, index, rows, cols, dev):
    """Keep-mask [rows][cols] (bool) of dense layer `index`'s dropout: Bernoulli(1 - p) from torch's device generator (Philox;
    graph-capture safe), or whatever the test hook `model._dropout_mask` returns."""
    if model._dropout_mask is not None:
        return model._dropout_mask(index, rows, cols, dev)
    return torch.rand((rows, cols), device=dev) >= model.drop_rate


class _DenseNetFn(Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        x = model._float_patches(x)       # uint8 patches: ToTensor (+ Normalize) as one pass (conv0's wgrad re-reads floats)
        N, _, P, _ = x.shape
        dev = x.device
        st = L.stream()
        training = model.training
        hs, sizes = model._geometry(P)
        mid = model.bn_size * model.growth_rate
        g = model.growth_rate
        conv0 = model.features.conv0
        c0 = conv0.out_channels
        tape = _Tape()
        tape.x, tape.N, tape.P, tape.hs, tape.sizes, tape.training = x, N, P, hs, sizes, training
        bufs = [torch.empty((N * s * s, c_total), device=dev, dtype=F32)
                for (_, _, _, c_total), s in zip(model._blocks, sizes)]
        tape.bufs = bufs
        w0 = conv0.weight.detach().contiguous()
        ld1 = bufs[0].shape[1]
        if model.small_inputs:
            L.call('gnx_conv_stem', L.ptr(x), L.ptr(w0), L.ptr(bufs[0]), ld1, N, 3, P, P, c0, 3, 3, 1, 1, st)
            tape.stem_out = tape.stats0 = None
        else:
            hp = (hs + 2 - 3) // 2 + 1
            tape.pool_idx = tape.stem_out = None
            rc = L.ERR_UNSUPPORTED
            if not training and c0 % 4 == 0 and ld1 % 4 == 0 and gammas_nonzero(model):
                # Running statistics (training.py:126 keeps f in eval mode): norm0's map is known before conv0 runs, so stem,
                # norm0, relu0 and pool0 are the eval forward's ONE kernel, which also records which window element won
                # (a byte per pooled element).  The backward routes by that index and reads mask and x_hat off the pooled
                # map: the 5.2 GB conv0 map of a 128-px array is never written.
                s0 = _bn(model.features.norm0, None, c0, N * hs * hs, False, dev, st)
                tape.pool_idx = torch.empty((N * hp * hp, c0), device=dev, dtype=torch.uint8)
                rc = L.query('gnx_conv_stem_bnrelu_maxpool_argmax', L.ptr(x), L.ptr(w0), L.ptr(bufs[0]), ld1,
                             tape.pool_idx.data_ptr(), N, 3, P, P, c0, 7, 7, 2, 3, L.ptr(s0[0]), L.ptr(s0[1]), st)
                if rc not in (0, L.ERR_UNSUPPORTED):
                    raise RuntimeError("gnx_conv_stem_bnrelu_maxpool_argmax failed (%d)" % rc)
            if rc == L.ERR_UNSUPPORTED:
                stem_out = torch.empty((N * hs * hs, c0), device=dev, dtype=F32)
                L.call('gnx_conv_stem', L.ptr(x), L.ptr(w0), L.ptr(stem_out), c0, N, 3, P, P, c0, 7, 7, 2, 3, st)
                s0 = _bn(model.features.norm0, L.ptr(stem_out), c0, N * hs * hs, training, dev, st)
                # pool0 records WHICH window element won (one byte per pooled element): its adjoint then routes by index, ties
                # exactly as torch's max_pool2d and without re-reading the conv0 map
                tape.pool_idx = None
                if c0 % 4 == 0:
                    tape.pool_idx = torch.empty((N * hp * hp, c0), device=dev, dtype=torch.uint8)
                    L.call('gnx_bnrelu_maxpool_argmax', L.ptr(stem_out), c0, L.ptr(bufs[0]), ld1, tape.pool_idx.data_ptr(), N,
                           c0, hs, hs, L.ptr(s0[0]), L.ptr(s0[1]), st)
                else:
                    L.call('gnx_bnrelu_maxpool', L.ptr(stem_out), c0, L.ptr(bufs[0]), ld1, N, c0, hs, hs, L.ptr(s0[0]),
                           L.ptr(s0[1]), st)
                tape.stem_out = stem_out
                if tape.pool_idx is not None and not training and ld1 % 4 == 0 and gammas_nonzero(model):
                    tape.stem_out = None  # running statistics: the backward works from the pooled map
            tape.stats0 = s0
        w2r = relayout_weights(model, 0, dev, st)            # conv2 weights tap-major, all layers, one launch
        tape.layers = []          # per block: list of (bott, stats1, stats2, bottleneck stored activated?)
        act_ok = (not training) and gammas_nonzero(model)
        w2u = model._winograd_conv2() if (act_ok and model.winograd and model.mfma == 'f32') else None
        split_ok = act_ok and model.mfma == 'f32' and mid == 128
        w1s = model._split_conv1() if (split_ok and model.split_conv1) else None
        w2s = model._split_conv2() if (split_ok and model.split_conv2 and g == 32) else None
        tape.trans = []           # per block: stats of the transition BN (or None)
        n_drop = 0
        for bi, ((c_in, layers, trans, c_total), s) in enumerate(zip(model._blocks, sizes)):
            buf = bufs[bi]
            M = N * s * s
            recs = []
            for li, layer in enumerate(layers):
                cin = c_in + li * g
                s1 = _bn(layer.norm1, L.ptr(buf), c_total, M, training, dev, st)
                bott = torch.empty((M, mid), device=dev, dtype=F32)
                # Eval statistics (training.py:126 puts f in eval mode even when it is trained): norm2's affine map is known
                # before conv1 runs, so the bottleneck is stored ACTIVATED - conv2 then runs its prologue-free LDS-DMA
                # form and its weight gradient needs no prologue; the BN adjoint recovers mask and
                # x_hat from the activated values (needs scale != 0, i.e. gamma != 0, checked once per layer).
                activated = act_ok
                if activated:
                    s2 = _bn(layer.norm2, None, mid, M, False, dev, st)
                    t0 = model._probe_begin()
                    rc = L.ERR_UNSUPPORTED
                    if w1s is not None:
                        # (opt-in, `model.split_conv1`) the same product on split bf16 operands: fp32-grade values, HBM-bound;
                        # the backward differentiates the same function on the fp32 instruction
                        rc = L.query('gnx_conv1x1_bnrelu_act_split', L.ptr(buf), c_total, w1s[layer].data_ptr(), L.ptr(bott), mid, M, cin,
                                     L.ptr(s1[0]), L.ptr(s1[1]), L.ptr(s2[0]), L.ptr(s2[1]), st)
                        if rc not in (0, L.ERR_UNSUPPORTED):
                            raise RuntimeError("gnx_conv1x1_bnrelu_act_split failed (%d)" % rc)
                    if rc == L.ERR_UNSUPPORTED:
                        L.call('gnx_conv1x1_bnrelu_act', L.ptr(buf), c_total, L.ptr(layer.conv1.weight), L.ptr(bott), mid, M,
                               mid, cin, L.ptr(s1[0]), L.ptr(s1[1]), L.ptr(s2[0]), L.ptr(s2[1]), st)
                    model._probe_mark('conv1x1', t0, 2 * M * cin * mid, 4 * M * (cin + mid))
                    # conv2 on the ready operand as in the eval forward: Winograd F(2,3) along x for maps of 8 x 8 and up
                    # (`model.winograd`), the direct form otherwise.  The backward is the adjoint of whichever ran (masks
                    # and x_hat come from the stored activations).  Against a direct-form reference the outputs differ at
                    # rounding level; only a pre-activation that a reference computes as EXACTLY 0 (closed-form nets with
                    # integer weights) can flip a ReLU mask of the next layer - such tests set `model.winograd = False`.
                    rc = L.ERR_UNSUPPORTED
                    t0 = model._probe_begin()
                    if w2s is not None:
                        rc = L.query('gnx_conv3x3_split', L.ptr(bott), mid, w2s[layer].data_ptr(), _cols(buf, cin), c_total, M, s, st)
                        if rc not in (0, L.ERR_UNSUPPORTED):
                            raise RuntimeError("gnx_conv3x3_split failed (%d)" % rc)
                    if rc == L.ERR_UNSUPPORTED and w2u is not None and s >= 8:
                        rc = L.query('gnx_conv3x3_winograd', L.ptr(bott), mid, L.ptr(w2u[layer]), _cols(buf, cin), c_total, M,
                                     g, mid, s, st)
                        if rc not in (0, L.ERR_UNSUPPORTED):
                            raise RuntimeError("gnx_conv3x3_winograd failed (%d)" % rc)
                    if rc == L.ERR_UNSUPPORTED:
                        L.call('gnx_conv3x3_bnrelu', L.ptr(bott), mid, L.ptr(w2r[layer.conv2.weight]), _cols(buf, cin), c_total,
                               M, g, mid, s, None, None, st)
                    model._probe_mark('conv3x3', t0, 2 * M * 9 * mid * g, 4 * M * (mid + g))
                else:
                    nws = L.query('gnx_conv1x1_workspace', M, mid, cin)       # small batches: K split over workgroups
                    ws1 = torch.empty(nws, device=dev, dtype=F32) if nws else None
                    L.call('gnx_conv1x1_bnrelu_ws', L.ptr(buf), c_total, L.ptr(layer.conv1.weight), L.ptr(bott), mid, M, mid,
                           cin, L.ptr(s1[0]), L.ptr(s1[1]), L.ptr(ws1), st)
                    s2 = _bn(layer.norm2, L.ptr(bott), mid, M, training, dev, st)
                    L.call('gnx_conv3x3_bnrelu', L.ptr(bott), mid, L.ptr(w2r[layer.conv2.weight]), _cols(buf, cin), c_total, M,
                           g, mid, s, L.ptr(s2[0]), L.ptr(s2[1]), st)
                keep = None
                if training and model.drop_rate > 0:
                    # F.dropout(new_features, p, training) (densenet.py:42-43): the layer's 32 new columns times a keep-mask
                    # / (1 - p), in place in the block buffer - later layers read the dropped features, as after the
                    # reference's concatenation; the backward applies the same factor to the columns' gradient
                    keep = _dropout_keep(model, n_drop, M, g, dev)
                    n_drop += 1
                    buf[:, cin:cin + g].mul_(keep.to(F32).mul_(1.0 / (1.0 - model.drop_rate)))
                recs.append((bott, s1, s2, activated, keep))
            tape.layers.append(recs)
            if trans is not None:
                nxt = bufs[bi + 1]
                so = s // 2
                stt = _bn(trans.norm, L.ptr(buf), c_total, M, training, dev, st)
                L.call('gnx_conv1x1_bnrelu', L.ptr(buf), c_total, L.ptr(trans.conv.weight), L.ptr(nxt), nxt.shape[1],
                       N * so * so, trans.conv.out_channels, c_total, L.ptr(stt[0]), L.ptr(stt[1]), 1, s, st)
                tape.trans.append(stt)
            else:
                tape.trans.append(None)
        s_last = sizes[-1]
        c_last = model.num_features
        sf = _bn(model.features.norm_final, L.ptr(bufs[-1]), bufs[-1].shape[1], N * s_last * s_last, training, dev, st)
        tape.statsf = sf
        feats = torch.empty((N, c_last), device=dev, dtype=F32)
        L.call('gnx_bnrelu_avgpool', L.ptr(bufs[-1]), bufs[-1].shape[1], L.ptr(feats), c_last, N, c_last,
               s_last * s_last, L.ptr(sf[0]), L.ptr(sf[1]), st)
        tape.feats = feats
        # backward re-reads the live conv / classifier weights: remember which values this forward used
        tape.versions = [(p, p._version, p.data_ptr()) for p in params]
        if training:
            model.invalidate_cache()      # running statistics were updated through raw pointers (no _version bump)
        ctx.tape, ctx.model = tape, model
        ctx.x_needs_grad = x.requires_grad
        if not model.classify:
            return feats.clone()
        nc = model.classifier.out_features
        out = torch.empty((N, nc), device=dev, dtype=F32)
        L.call('gnx_gemm_f32', L.ptr(feats), c_last, 0, L.ptr(model.classifier.weight), c_last, 0,
               L.ptr(model.classifier.bias), L.ptr(out), nc, N, nc, c_last, 0, st)
        return out

    @staticmethod
    def backward(ctx, dout):
        model, tape = ctx.model, ctx.tape
        if tape is None:
            raise RuntimeError("gridnext_amd.DenseNet: the tape of this forward was already consumed (a second backward / "
                               "retain_graph=True is not supported: run the forward again)")
        for p, ver, addr in tape.versions:
            if p._version != ver or p.data_ptr() != addr:
                raise RuntimeError("gridnext_amd.DenseNet: a parameter was modified between forward and backward "
                                   "(optimizer.step() or load_state_dict before loss.backward()); its gradient would be "
                                   "computed from the new value")
        if ctx.x_needs_grad:
            raise NotImplementedError("gradient with respect to the input patches is not part of the GridNext path")
        dout = dout.contiguous()
        dev = dout.device
        st = L.stream()
        N, P, hs, sizes, training = tape.N, tape.P, tape.hs, tape.sizes, tape.training
        g = model.growth_rate
        mid = model.bn_size * g
        grads = {}

        def want(p):
            return p is not None and p.requires_grad

        def new_like(p):
            t = torch.empty_like(p, memory_format=torch.contiguous_format)
            grads[p] = t
            return t

        def bn_bwd(bn, stats, dy_ptr, lddy, x_ptr, ldx, dx_ptr, lddx, M, C, dx_acc, relu=1):
            dg = new_like(bn.weight) if want(bn.weight) else None
            db = new_like(bn.bias) if want(bn.bias) else None
            ws = torch.empty(L.query('gnx_bn_workspace', M, C), device=dev, dtype=F32)
            sync = _bn_sync(bn, torch.device(dev))
            L.call('gnx_bn_relu_bwd_sync', dy_ptr, lddy, x_ptr, ldx, dx_ptr, lddx, M, C, L.ptr(stats[0]), L.ptr(stats[1]),
                   L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(dg), L.ptr(db), relu, 1 if training else 0, 0, dx_acc,
                   L.ptr(ws), sync.data_ptr() if sync is not None else None, st)

        def wgrad(w, dy_ptr, lddy, x_ptr, ldx, stats, M, Nn, K, S, taps, pool):
            if not want(w):
                return
            dw = new_like(w)
            sc, sh = (L.ptr(stats[0]), L.ptr(stats[1])) if stats is not None else (None, None)
            if model.split_wgrad and taps == 1 and not pool and Nn % 4 == 0 and K % 4 == 0:
                # (opt-in) the same contraction on split bf16 operands: HBM-bound instead of bound by the fp32 matrix instruction
                t0 = model._probe_begin()
                ws = torch.empty(L.query('gnx_wgrad1x1_split_workspace', M, Nn, K), device=dev, dtype=F32)
                rc = L.query('gnx_wgrad1x1_split', dy_ptr, lddy, x_ptr, ldx, sc, sh, L.ptr(dw), L.ptr(ws), M, Nn, K, 0, L.stream())
                if rc == 0:
                    model._probe_mark('wgrad1x1', t0, 2 * M * Nn * K, 4 * M * (Nn + K))
                    return
                if rc != L.ERR_UNSUPPORTED:
                    raise RuntimeError("gnx_wgrad1x1_split failed (%d)" % rc)
            if model.split_wgrad and taps == 9 and stats is None and Nn == 32 and K == 128 and S >= 8:
                # (opt-in) conv2's weight gradient from the activated bottleneck, same arithmetic (4 x 4 maps: the fp32 kernel is faster)
                t0 = model._probe_begin()
                ws = torch.empty(L.query('gnx_wgrad3x3_split_workspace', M), device=dev, dtype=F32)
                rc = L.query('gnx_wgrad3x3_split', dy_ptr, lddy, x_ptr, ldx, L.ptr(dw), L.ptr(ws), M, S, 0, L.stream())
                if rc == 0:
                    model._probe_mark('wgrad3x3', t0, 2 * M * 9 * Nn * K, 4 * M * (Nn + K))
                    return
                if rc != L.ERR_UNSUPPORTED:
                    raise RuntimeError("gnx_wgrad3x3_split failed (%d)" % rc)
            ws = torch.empty(L.query('gnx_wgrad_workspace', M, Nn, K, taps), device=dev, dtype=F32)
            t0 = model._probe_begin()
            L.call('gnx_wgrad_bnrelu', dy_ptr, lddy, x_ptr, ldx, sc, sh, L.ptr(dw), L.ptr(ws), M, Nn, K, S, taps, pool, 0,
                   L.stream())
            model._probe_mark('wgrad3x3' if taps == 9 else ('wgrad_trans' if pool else 'wgrad1x1'), t0, 2 * M * taps * Nn * K,
                              4 * M * (Nn + K * (4 if pool else 1)))

        # Under hipGraph capture (the spot loop's small-batch steps, graphs.py) a dense block's weight gradients are DEFERRED to
        # the end of the block and launched on a side stream as ONE parallel branch of the step graph: they are off the critical
        # path (the data-gradient chain), and at a batch of 32 patches every kernel leaves most of the chip idle, so the branch
        # runs beside the next block's chain.  One fork per block, one join at the end: 2 514 -> 2 577 spots/s in the same run
        # (a fork / join per LAYER cost more than it gave: 2 369 against 2 505; branches of 2, 4 or 8 layers: the same +2 %).
        # Each layer then keeps its own gradient-of-bottleneck buffer until the join.
        side = None
        if torch.cuda.is_current_stream_capturing():
            side = model.__dict__.get('_wgrad_stream')
            if side is None or side.device != dev:
                side = model.__dict__['_wgrad_stream'] = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        deferred, held, side_done = [], [], []

        def wgrad_batch(calls):
            """The deferred weight gradients of one dense block.  Calls that are plain (w, dy_ptr, ...) argument tuples of one kind
            go out as ONE launch per 24 layers plus one batched reduce (gnx_wgrad_bnrelu_batch: the same kernel bodies, slab
            layout and reduce order - bit-identical to the single calls); whatever that entry point does not take, and the
            callables, are made one by one."""
            for taps in (9, 1):
                items = [c for c in calls if isinstance(c, tuple) and c[10] == taps and c[11] == 0 and want(c[0])]
                if taps == 1:                                       # (K <= 128 runs another kernel: those go out singly)
                    items = [c for c in items if c[8] > 128 and c[8] % 4 == 0 and c[7] % 128 == 0]
                if len(items) < 2:
                    continue
                arr = (_WgradItem * len(items))()
                keep = []
                for a, (w, dy_ptr, lddy, x_ptr, ldx, stats, M_, Nn, K_, S_, _t, _p) in zip(arr, items):
                    dw = torch.empty_like(w, memory_format=torch.contiguous_format)
                    ws = torch.empty(L.query('gnx_wgrad_workspace', M_, Nn, K_, taps), device=dev, dtype=F32)
                    keep.append((dw, ws))
                    a.dY, a.lddy, a.X, a.ldx = dy_ptr, lddy, x_ptr, ldx
                    a.scale, a.shift = (L.ptr(stats[0]), L.ptr(stats[1])) if stats is not None else (None, None)
                    a.dW, a.workspace, a.M, a.N, a.K, a.S, a.accumulate = L.ptr(dw), L.ptr(ws), M_, Nn, K_, S_, 0
                rc = L.query('gnx_wgrad_bnrelu_batch', ctypes.addressof(arr), len(items), taps, L.stream())
                if rc == 0:
                    for (w, *_), (dw, _ws) in zip(items, keep):
                        grads[w] = dw
                    done_ids = {id(c) for c in items}
                    calls = [c for c in calls if id(c) not in done_ids]
                elif rc != L.ERR_UNSUPPORTED:
                    raise RuntimeError("gnx_wgrad_bnrelu_batch failed (%d)" % rc)
            for c in calls:
                if isinstance(c, tuple):
                    wgrad(*c)
                else:
                    c()

        def flush_deferred():
            if not deferred:
                return
            ev = torch.cuda.Event()
            ev.record(cur)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                wgrad_batch(list(deferred))
            deferred.clear()
            done = torch.cuda.Event()
            done.record(side)
            side_done.append(done)

        # data-parallel runs: this node's gradients leave block by block, each bucket's all-reduce overlapping the rest of the
        # backward (distributed.BackwardReducer); what autograd receives is already the average over ranks
        from . import distributed as gdist
        reducer = gdist.BackwardReducer() if gdist.BackwardReducer.wanted() else None
        if gdist.is_active():
            gdist.note_backward(reducer is not None)     # (all backwards of one optimizer step must deliver alike)
        sent = set()

        def send_bucket():
            if reducer is None:
                return
            ps = [p for p in grads if id(p) not in sent]
            sent.update(id(p) for p in ps)
            reducer.bucket([grads[p] for p in ps], ps)

        w2b = relayout_weights(model, 1, dev, st)            # conv2 weights for the data gradient, all layers, one launch
        w1ts = relayout_weights(model, 2, dev, st)           # conv1 / transition weights transposed, one launch
        # ---- classifier
        c_last = model.num_features
        if model.classify:
            cls = model.classifier
            nc = cls.out_features
            dfeats = torch.empty((N, c_last), device=dev, dtype=F32)
            L.call('gnx_gemm_f32', L.ptr(dout), nc, 0, L.ptr(cls.weight), c_last, 1, None, L.ptr(dfeats), c_last,
                   N, c_last, nc, 0, st)
            if want(cls.weight):
                dw = new_like(cls.weight)
                L.call('gnx_gemm_f32', L.ptr(dout), nc, 1, L.ptr(tape.feats), c_last, 1, None, L.ptr(dw), c_last,
                       nc, c_last, N, 0, st)
            if want(cls.bias):
                db = new_like(cls.bias)
                ws = torch.empty(L.query('gnx_bn_workspace', N, nc), device=dev, dtype=F32)
                L.call('gnx_colsum', L.ptr(dout), nc, N, nc, L.ptr(db), 0, L.ptr(ws), st)
        else:
            dfeats = dout

        # ---- tail: norm_final -> relu -> global average
        bufs = tape.bufs
        dbufs = [None] * len(bufs)
        s_last = sizes[-1]
        M_last = N * s_last * s_last
        dY = torch.empty((M_last, c_last), device=dev, dtype=F32)
        L.call('gnx_rows_broadcast', L.ptr(dfeats), c_last, L.ptr(dY), c_last, N, c_last, s_last * s_last,
               1.0 / (s_last * s_last), st)
        dbufs[-1] = torch.empty_like(bufs[-1])
        bn_bwd(model.features.norm_final, tape.statsf, L.ptr(dY), c_last, L.ptr(bufs[-1]), bufs[-1].shape[1],
               L.ptr(dbufs[-1]), bufs[-1].shape[1], M_last, c_last, 0)
        del dY

        # ---- dense blocks, last to first
        for bi in range(len(model._blocks) - 1, -1, -1):
            c_in, layers, trans, c_total = model._blocks[bi]
            s = sizes[bi]
            M = N * s * s
            buf, dbuf = bufs[bi], dbufs[bi]
            tA = torch.empty((M, mid), device=dev, dtype=F32)
            tB = torch.empty((M, mid), device=dev, dtype=F32)
            tC = torch.empty((M, c_total), device=dev, dtype=F32)
            for li in range(len(layers) - 1, -1, -1):
                layer = layers[li]
                bott, s1, s2, activated, keep = tape.layers[bi][li]
                cin = c_in + li * g
                if keep is not None:                               # dropout's adjoint on this layer's gradient columns
                    dbuf[:, cin:cin + g].mul_(keep.to(F32).mul_(1.0 / (1.0 - model.drop_rate)))
                dy2 = _cols(dbuf, cin)
                if side is not None:
                    tB = torch.empty((M, mid), device=dev, dtype=F32)      # this layer's own: read by its deferred wgrad
                    held.append((bott, s1, s2, tB, buf, dbuf))
                # conv2: weight gradient (no prologue when the bottleneck was stored activated), then data gradient
                # (adjoint conv with flipped taps)
                w2_args = (layer.conv2.weight, dy2, c_total, L.ptr(bott), mid, None if activated else s2, M, g, mid, s, 9, 0)
                if side is not None:
                    deferred.append(w2_args)
                else:
                    wgrad(*w2_args)
                wb = w2b[layer.conv2.weight]
                # conv2's data gradient and norm2 -> relu2's adjoint: ONE kernel where the bottleneck was stored activated
                # (eval statistics) and the shape is the LDS-DMA kernel's; otherwise the product, then the adjoint pass
                rc = L.ERR_UNSUPPORTED
                if activated:
                    bn2 = layer.norm2
                    dg2 = new_like(bn2.weight) if want(bn2.weight) else None
                    db2 = new_like(bn2.bias) if want(bn2.bias) else None
                    ws2 = torch.empty(L.query('gnx_conv3x3_dgrad_bn_workspace', M, mid), device=dev, dtype=F32)
                    t0 = model._probe_begin()
                    rc = L.query('gnx_conv3x3_dgrad_bnrelu_bwd', dy2, c_total, L.ptr(wb), L.ptr(bott), mid, L.ptr(tB), mid, M,
                                 mid, g, s, L.ptr(s2[0]), L.ptr(s2[1]), L.ptr(s2[2]), L.ptr(s2[3]), L.ptr(dg2), L.ptr(db2), 0,
                                 L.ptr(ws2), st)
                    if rc not in (0, L.ERR_UNSUPPORTED):
                        raise RuntimeError("gnx_conv3x3_dgrad_bnrelu_bwd failed (%d)" % rc)
                    if rc == 0:
                        model._probe_mark('dgrad3x3_bn2', t0, 2 * M * 9 * mid * g, 4 * M * (g + 2 * mid))
                if rc == L.ERR_UNSUPPORTED:
                    t0 = model._probe_begin()
                    L.call('gnx_conv3x3_bnrelu', dy2, c_total, L.ptr(wb), L.ptr(tA), mid, M, mid, g, s, None, None, st)
                    t0 = model._probe_mark('dgrad3x3', t0, 2 * M * 9 * mid * g, 4 * M * (g + mid))
                    # norm2 + relu2
                    bn_bwd(layer.norm2, s2, L.ptr(tA), mid, L.ptr(bott), mid, L.ptr(tB), mid, M, mid, 0,
                           relu=2 if activated else 1)
                    model._probe_mark('bn2_bwd', t0, 0, 4 * M * 3 * mid)
                # conv1
                w1_args = (layer.conv1.weight, L.ptr(tB), mid, L.ptr(buf), c_total, s1, M, mid, cin, s, 1, 0)
                w1t = w1ts[layer.conv1.weight]
                # Optional (`model.fused_conv1_backward = True`): data gradient + norm1/relu1 adjoint into the block gradient AND
                # the weight gradient from the same staged tiles, ONE pass over dB, X and G (gnx_conv1x1_dgrad_wgrad_bnrelu_bwd).
                # Not the default: with both products the fp32 pass is bound by the matrix pipe at ~100 TFLOP/s (93 ms per
                # 128-px array) - the same time as the two separate passes it replaces (50 + 42 ms); on the
                # fp16 path, where the matrix work is 16x cheaper, the same fusion is the default (densenet_train_f16)
                rc = L.ERR_UNSUPPORTED
                done_w1 = False
                if not training and side is None and mid == 128 and cin % 32 == 0 and M % 32 == 0 and want(layer.conv1.weight) and \
                        model.__dict__.get('fused_conv1_backward', False):
                    bn1 = layer.norm1
                    dg = new_like(bn1.weight) if want(bn1.weight) else None
                    db = new_like(bn1.bias) if want(bn1.bias) else None
                    ws = torch.empty(L.query('gnx_conv1x1_dgrad_wgrad_workspace', M, cin), device=dev, dtype=F32)
                    dw1 = new_like(layer.conv1.weight)
                    t0 = model._probe_begin()
                    rc = L.query('gnx_conv1x1_dgrad_wgrad_bnrelu_bwd', L.ptr(tB), mid, L.ptr(w1t), L.ptr(buf), c_total, L.ptr(dbuf),
                                 c_total, M, cin, L.ptr(s1[0]), L.ptr(s1[1]), L.ptr(s1[2]), L.ptr(s1[3]), L.ptr(dg), L.ptr(db),
                                 L.ptr(dw1), L.ptr(ws), 0, st)
                    if rc not in (0, L.ERR_UNSUPPORTED):
                        raise RuntimeError("gnx_conv1x1_dgrad_wgrad_bnrelu_bwd failed (%d)" % rc)
                    if rc == 0:
                        done_w1 = True
                        model._probe_mark('dgrad_wgrad1x1_bn1', t0, 4 * M * cin * mid, 4 * M * (mid + 3 * cin))
                if not done_w1:
                    if side is not None:
                        deferred.append(w1_args)
                    else:
                        wgrad(*w1_args)
                # conv1's data gradient + norm1/relu1 backward, accumulated into the block-buffer gradient: one kernel where
                # the statistics are the running ones and the tiles are whole (3 passes over [M][cin] instead of 5)
                if not done_w1 and not training:
                    bn1 = layer.norm1
                    dg = new_like(bn1.weight) if want(bn1.weight) else None
                    db = new_like(bn1.bias) if want(bn1.bias) else None
                    ws = torch.empty(L.query('gnx_conv1x1_dgrad_bn_workspace', M, cin), device=dev, dtype=F32)
                    t0 = model._probe_begin()
                    rc = L.query('gnx_conv1x1_dgrad_bnrelu_bwd', L.ptr(tB), mid, L.ptr(w1t), L.ptr(buf), c_total,
                                 L.ptr(dbuf), c_total, M, cin, mid, L.ptr(s1[0]), L.ptr(s1[1]), L.ptr(s1[2]), L.ptr(s1[3]),
                                 L.ptr(dg), L.ptr(db), 0, L.ptr(ws), st)
                    if rc not in (0, L.ERR_UNSUPPORTED):
                        raise RuntimeError("gnx_conv1x1_dgrad_bnrelu_bwd failed (%d)" % rc)
                    if rc == 0:
                        model._probe_mark('dgrad1x1_bn1', t0, 2 * M * cin * mid, 4 * M * (mid + 3 * cin))
                if rc == L.ERR_UNSUPPORTED:
                    L.call('gnx_conv1x1_bnrelu', L.ptr(tB), mid, L.ptr(w1t), L.ptr(tC), c_total, M, cin, mid, None, None,
                           0, 0, st)
                    bn_bwd(layer.norm1, s1, L.ptr(tC), c_total, L.ptr(buf), c_total, L.ptr(dbuf), c_total, M, cin, 1)
                tape.layers[bi][li] = None
            if side is not None:
                flush_deferred()                                   # this block's weight gradients: one branch on the side stream
            del tA, tB, tC
            if bi > 0:
                # transition bi-1 -> bi : dT is columns [0, c_in) of this block's gradient
                p_c_in, p_layers, p_trans, p_total = model._blocks[bi - 1]
                ps = sizes[bi - 1]
                Mp = N * ps * ps
                stt = tape.trans[bi - 1]
                c_out = p_trans.conv.out_channels
                # weight gradient: with eval statistics the pooled, activated input is built once (one 16-B pass) and the plain
                # 1x1 weight-gradient kernel runs on a quarter of the rows; the in-kernel pooling form otherwise
                rc = L.ERR_UNSUPPORTED
                if want(p_trans.conv.weight) and not training:
                    pooled = torch.empty((M, p_total), device=dev, dtype=F32)
                    rc = L.query('gnx_bnrelu_avgpool2', L.ptr(bufs[bi - 1]), p_total, L.ptr(pooled), p_total, N, p_total, ps,
                                 L.ptr(stt[0]), L.ptr(stt[1]), st)
                    if rc == 0:
                        wgrad(p_trans.conv.weight, L.ptr(dbuf), c_total, L.ptr(pooled), p_total, None, M, c_out, p_total, 0, 1, 0)
                    elif rc != L.ERR_UNSUPPORTED:
                        raise RuntimeError("gnx_bnrelu_avgpool2 failed (%d)" % rc)
                    del pooled
                if rc == L.ERR_UNSUPPORTED:
                    wgrad(p_trans.conv.weight, L.ptr(dbuf), c_total, L.ptr(bufs[bi - 1]), p_total, stt, M, c_out, p_total,
                          ps, 1, 1)
                wt = w1ts[p_trans.conv.weight]
                dPool = torch.empty((M, p_total), device=dev, dtype=F32)
                L.call('gnx_conv1x1_bnrelu', L.ptr(dbuf), c_total, L.ptr(wt), L.ptr(dPool), p_total, M, p_total, c_out,
                       None, None, 0, 0, st)
                dbufs[bi - 1] = torch.empty_like(bufs[bi - 1])
                # norm -> relu adjoint straight from the POOLED gradient (eval statistics): the unpooled map - a full-size
                # write and read - is never built
                rc = L.ERR_UNSUPPORTED
                if not training:
                    bnp = p_trans.norm
                    dgp = new_like(bnp.weight) if want(bnp.weight) else None
                    dbp = new_like(bnp.bias) if want(bnp.bias) else None
                    wsp = torch.empty(L.query('gnx_bn_workspace', Mp, p_total), device=dev, dtype=F32)
                    rc = L.query('gnx_bn_relu_bwd_pooled', L.ptr(dPool), p_total, L.ptr(bufs[bi - 1]), p_total,
                                 L.ptr(dbufs[bi - 1]), p_total, N, ps, p_total, L.ptr(stt[0]), L.ptr(stt[1]), L.ptr(stt[2]),
                                 L.ptr(stt[3]), L.ptr(dgp), L.ptr(dbp), 0, L.ptr(wsp), st)
                    if rc not in (0, L.ERR_UNSUPPORTED):
                        raise RuntimeError("gnx_bn_relu_bwd_pooled failed (%d)" % rc)
                if rc == L.ERR_UNSUPPORTED:
                    dAct = torch.empty((Mp, p_total), device=dev, dtype=F32)
                    L.call('gnx_avgpool2_bwd', L.ptr(dPool), p_total, L.ptr(dAct), p_total, N, p_total, ps, st)
                    bn_bwd(p_trans.norm, stt, L.ptr(dAct), p_total, L.ptr(bufs[bi - 1]), p_total, L.ptr(dbufs[bi - 1]),
                           p_total, Mp, p_total, 0)
                    del dAct
                del dPool
                dbufs[bi] = None
                bufs[bi] = None
            send_bucket()                                          # this block (+ the transition below it): final

        # ---- stem
        conv0 = model.features.conv0
        c0 = conv0.out_channels
        c_total1 = bufs[0].shape[1]
        if want(conv0.weight):
            dw0 = new_like(conv0.weight)
            if model.small_inputs:
                ws = torch.empty(L.query('gnx_conv0_wgrad_workspace', N, P, P, c0, 3, 3, 1, 1), device=dev, dtype=F32)
                L.call('gnx_conv0_wgrad', L.ptr(tape.x), L.ptr(dbufs[0]), c_total1, L.ptr(dw0), L.ptr(ws), N, P, P, c0,
                       3, 3, 1, 1, 0, st)
        if not model.small_inputs:
            M0 = N * hs * hs
            need = want(conv0.weight) or want(model.features.norm0.weight) or want(model.features.norm0.bias)
            if need:
                s0 = tape.stats0
                dS = torch.empty((M0, c0), device=dev, dtype=F32)
                rc = L.ERR_UNSUPPORTED
                if tape.pool_idx is not None and not training and gammas_nonzero(model):
                    # running statistics: pool0's adjoint carries norm0 -> relu0 with it (mask and x_hat from the pooled
                    # activated map in block 1's buffer), the conv0 map is not read again
                    rc = L.query('gnx_maxpool_bwd_argmax_bnrelu', tape.pool_idx.data_ptr(), L.ptr(dbufs[0]), c_total1,
                                 L.ptr(bufs[0]), c_total1, L.ptr(s0[0]), L.ptr(dS), c0, N, c0, hs, hs, st)
                    if rc not in (0, L.ERR_UNSUPPORTED):
                        raise RuntimeError("gnx_maxpool_bwd_argmax_bnrelu failed (%d)" % rc)
                    if rc == 0:
                        hp = (hs + 2 - 3) // 2 + 1
                        bn_bwd(model.features.norm0, s0, L.ptr(dbufs[0]), c_total1, L.ptr(bufs[0]), c_total1, None, c0,
                               N * hp * hp, c0, 0, relu=2)
                if rc == L.ERR_UNSUPPORTED:
                    if tape.stem_out is None:
                        raise RuntimeError("the conv0 map was not kept and gnx_maxpool_bwd_argmax_bnrelu refused the shapes")
                    dAct = torch.empty((M0, c0), device=dev, dtype=F32)
                    if tape.pool_idx is not None:
                        L.call('gnx_maxpool_bwd_argmax', tape.pool_idx.data_ptr(), L.ptr(dbufs[0]), c_total1, L.ptr(dAct), c0,
                               N, c0, hs, hs, st)
                    else:
                        L.call('gnx_maxpool_bwd', L.ptr(tape.stem_out), c0, L.ptr(bufs[0]), c_total1, L.ptr(dbufs[0]),
                               c_total1, L.ptr(dAct), c0, N, c0, hs, hs, L.ptr(s0[0]), L.ptr(s0[1]), st)
                    bn_bwd(model.features.norm0, s0, L.ptr(dAct), c0, L.ptr(tape.stem_out), c0, L.ptr(dS), c0, M0, c0, 0)
                    del dAct
                if want(conv0.weight):
                    ws = torch.empty(L.query('gnx_conv0_wgrad_workspace', N, P, P, c0, 7, 7, 2, 3), device=dev, dtype=F32)
                    L.call('gnx_conv0_wgrad', L.ptr(tape.x), L.ptr(dS), c0, L.ptr(grads[conv0.weight]), L.ptr(ws), N, P,
                           P, c0, 7, 7, 2, 3, 0, st)
        for ev in side_done:
            cur.wait_event(ev)                                     # join the weight-gradient branches
        held.clear()
        if reducer is not None:
            send_bucket()                                          # the stem's
            reducer.finish()
        ctx.tape = None
        out = [None, None]
        for p in model.parameters():
            out.append(grads.get(p))
        return tuple(out)


def _taped(model, x):
    """The autograd node that runs this call: the fp16-MFMA tape + backward (densenet_train_f16) where `DenseNet.mfma = 'f16'`
    and the shapes / mode are the ones it takes, the fp32 one otherwise."""
    if model.mfma == 'f16':
        from . import densenet_train_f16 as f16
        if f16.eligible(model, x):
            return f16._DenseNetF16Fn
    return _DenseNetFn


class _RecomputeFn(Function):
    """The same forward WITHOUT a tape; the backward runs the taped forward again on the saved input chunk and then its
    backward - what `cp.checkpoint` around a chunk (/root/reference/gridnext/gridnet_models.py:88-104) and `efficient=True`
    (/root/reference/gridnext/densenet.py:12-18, :36-40) do in the reference.  Memory: only the chunk's INPUT lives between
    forward and backward; one tape exists at a time, inside backward.
    BatchNorm running statistics (train mode only): updated exactly ONCE per chunk - the recompute starts from the
    statistics the first forward started from, so it reproduces that forward bit for bit (the reference's reentrant
    checkpoint runs the update twice).  Parameter gradients are accumulated straight into `.grad` by the inner backward, as
    a reentrant checkpoint does; nothing is returned for them."""

    @staticmethod
    def forward(ctx, model, x, *params):
        if x.requires_grad:
            raise NotImplementedError("gradient with respect to the input patches is not part of the GridNext path")
        ctx.model, ctx.x = model, x
        ctx.versions = [(p, p._version, p.data_ptr()) for p in params]
        ctx.bn_state = None
        # train-mode dropout: the recompute must draw the masks the first forward drew
        ctx.rng = torch.cuda.get_rng_state(x.device) if (model.training and model.drop_rate > 0) else None
        if model.training:
            ctx.bn_state = [(m, m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone())
                            for m in model._bn_modules()]
        with torch.no_grad():
            out = _taped(model, x).apply(model, x, *params)     # its tape dies with this call
        return out

    @staticmethod
    def backward(ctx, dout):
        model, x = ctx.model, ctx.x
        for p, ver, addr in ctx.versions:
            if p._version != ver or p.data_ptr() != addr:
                raise RuntimeError("gridnext_amd.DenseNet: a parameter was modified between forward and backward "
                                   "(optimizer.step() or load_state_dict before loss.backward()); the recomputed forward "
                                   "would differ from the one whose output the loss was computed from")
        after = None
        if ctx.bn_state is not None:
            after = [(m, m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone())
                     for m, _, _, _ in ctx.bn_state]
            for m, rm, rv, nb in ctx.bn_state:
                m.running_mean.copy_(rm)
                m.running_var.copy_(rv)
                m.num_batches_tracked.copy_(nb)
            model.invalidate_cache()
        rng_now = None
        if ctx.rng is not None:
            rng_now = torch.cuda.get_rng_state(x.device)
            torch.cuda.set_rng_state(ctx.rng, x.device)
        with torch.enable_grad():
            out = _taped(model, x).apply(model, x.detach(), *list(model.parameters()))
        if rng_now is not None:
            torch.cuda.set_rng_state(rng_now, x.device)
        if after is not None:                                    # later chunks may have moved the statistics on: keep theirs
            for m, rm, rv, nb in after:
                m.running_mean.copy_(rm)
                m.running_var.copy_(rv)
                m.num_batches_tracked.copy_(nb)
            model.invalidate_cache()
        torch.autograd.backward(out, dout)
        ctx.x = ctx.bn_state = None
        return (None, None) + (None,) * len(list(model.parameters()))


def tape_bytes_per_spot(model, P):
    """HBM one spot holds on the tape of the f-trained forward, plus its share of the backward's scratch (bytes): block
    buffers, one bottleneck per dense layer, the largest block's gradient buffers, the float patch."""
    hs, sizes = model._geometry(P)
    mid = model.bn_size * model.growth_rate
    total, biggest = 0, 0
    for (c_in, layers, trans, c_total), s in zip(model._blocks, sizes):
        block = s * s * c_total
        total += block + len(layers) * s * s * mid
        biggest = max(biggest, 2 * block + 2 * s * s * mid)
    return 4 * (total + biggest + 3 * P * P)


def densenet_autograd(model, x):
    """Differentiable DenseNet forward (training-mode BN when model.training, running-stat BN otherwise).
    Memory control (the reference's `efficient=True` and chunk checkpointing): with `model.efficient`, or when the tape of
    the whole batch would exceed `model.tape_budget` bytes under running statistics (where chunks are independent), the
    batch goes through in chunks whose forward keeps no tape and whose backward recomputes it (`_RecomputeFn`)."""
    params = list(model.parameters())
    n = x.shape[0]
    fn = _taped(model, x)
    if fn is _DenseNetFn:
        per = tape_bytes_per_spot(model, x.shape[2])
    else:
        from .densenet_train_f16 import tape_bytes_per_spot as per_f16
        per = per_f16(model, x.shape[2])
    chunk = n
    budget = getattr(model, 'tape_budget', None)
    if not model.training and budget and per * n > budget:
        chunk = max(8, int(budget // per) // 8 * 8)
    if chunk >= n:
        if getattr(model, 'efficient', False):
            return _RecomputeFn.apply(model, x, *params)
        return fn.apply(model, x, *params)
    return torch.cat([_RecomputeFn.apply(model, x.narrow(0, s0, min(chunk, n - s0)), *params) for s0 in range(0, n, chunk)], 0)


def densenet_recompute(model, x):
    """One chunk, forward without tape, recompute in backward (GridNet.atonce_patch_limit on the gradient path)."""
    if not x.is_cuda:
        raise RuntimeError("gridnext_amd.DenseNet runs on a HIP device only (input is on %s)" % x.device)
    return _RecomputeFn.apply(model, x.contiguous(), *list(model.parameters()))
