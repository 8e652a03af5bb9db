"""DenseNet-BC image spot classifier, MI355X-native.

Drop-in for `gridnext.densenet.DenseNet` (/root/reference/gridnext/densenet.py:78-159): same constructor
(:93-95), same parameter/buffer names (state_dict keys `features.conv0.weight`,
`features.denseblock{b}.denselayer{l}.{norm1,conv1,norm2,conv2}.*`, `features.transition{t}.{norm,conv}.*`,
`features.norm_final.*`, `classifier.*`), same initialisation (:141-150), same output.

The module tree below holds PARAMETERS ONLY (stock torch containers, so `.to()`, `.train()`, `state_dict()`
and optimizers behave as for the reference).  `forward` does not call them: it drives the hand-written gfx950
kernels of csrc/conv1x1.hip, conv3x3.hip and stem_pool.hip through the C ABI:
  * activations are channels-last matrices [spots*S*S, C]; a dense block is ONE buffer, each layer writes its
    `growth_rate` columns in place, so the reference's `torch.cat` (:14, :75) never happens;
  * BN+ReLU are folded into the operand load of the following conv; the transition averages 2x2 first;
  * a whole array's spots go through each kernel together (chunks only when activations would exceed ~40 GB,
    or when the caller sets `atonce` / GridNet.atonce_patch_limit).
`efficient=True` (checkpointing, :36-40): on the gradient path the forward keeps no tape and the backward recomputes it
(densenet_train._RecomputeFn); independently of the flag, a batch whose tape would exceed `tape_budget` bytes goes through in
recomputed chunks when BatchNorm runs on running statistics (train_gridwise always: training.py:126).
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as GF

F32 = torch.float32


class _DenseLayer(nn.Module):
    def __init__(self, c_in, growth_rate, bn_size, drop_rate):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(c_in)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(c_in, bn_size * growth_rate, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth_rate)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1, bias=False)
        self.drop_rate = drop_rate


class _Transition(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.norm = nn.BatchNorm2d(c_in)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(c_in, c_out, kernel_size=1, stride=1, bias=False)
        self.pool = nn.AvgPool2d(kernel_size=2, stride=2)


class DenseNet(nn.Module):
    def __init__(self, growth_rate=12, block_config=(16, 16, 16), compression=0.5,
                 num_init_features=24, bn_size=4, drop_rate=0,
                 num_classes=10, small_inputs=True, efficient=False, classify=True):
        super().__init__()
        assert 0 < compression <= 1, 'compression of densenet should be between 0 and 1'
        self.drop_rate = float(drop_rate or 0)      # dropout after conv2 (:42-43): a keep-mask on the layer's 32 new columns in
                                                    # training mode (densenet_train), identity otherwise
        self.growth_rate, self.block_config = growth_rate, tuple(block_config)
        self.bn_size, self.small_inputs, self.classify = bn_size, small_inputs, classify
        self.efficient = bool(efficient)      # gradient path: no tape in the forward, recompute in the backward (:12-18, :36-40)
        self.tape_budget = 150 * 1024 ** 3    # bytes of tape one backward may hold (eval-statistics gradient path): beyond
                                              # it the batch is cut into recomputed chunks (a 256-px array: 3 chunks)
        self.atonce = None          # spots per chunk in eval mode (None = auto)
        self.mfma = 'f32'           # 'f16': fp16 matrix-core operands in the eval forward (BASELINE config 5)
        self.split_conv1 = False    # eval forward, fp32 path: conv1 on SPLIT bf16 operands (three 16-bit matrix instructions per
                                    # product, fp32 tensors and accumulation: csrc/conv1x1_split.hip; opt-in, fp32-grade results)
        self.split_conv2 = False    # ... and conv2 as nine shifted products of split bf16 operands (csrc/conv3x3_split.hip)
        self.split_wgrad = False    # fp32 gradient path: conv1's weight gradient on split bf16 operands (csrc/wgrad_split.hip)
        self.winograd = True        # eval forward: conv2 as Winograd F(2,3) along x where the shape allows (fp32 path;
                                    # same arithmetic type, 1.5x fewer matrix operations, rounding-level differences)
        self.f16_buffers = True     # mfma = 'f16' only: the block buffers themselves in fp16 where the shapes allow
        self.f16_stem = True        # ... and, with fp16 block buffers, conv0's matrix operands in fp16 too
        self.f16_fused = True       # ... and every dense layer as ONE kernel, the bottleneck in LDS only (gnx_dense_layer_f16)
        self.f16_fused_transitions = True   # ... and every transition as ONE kernel, the pooled operand in LDS only (gnx_transition_f16)
        self.f16_fused_conv2_backward = True    # fp16 gradient path: conv2's data + weight gradient in one pass (gnx_conv3x3_bwd_f16_lb)
        self.input_norm = None      # (mean[3], std[3]) of a torchvision Normalize to apply to UINT8 input patches after the
                                    # u8 / 255 of ToTensor (fused into the stem's operand load); float inputs are taken as
                                    # already transformed by the dataset, as in the reference

        feats = OrderedDict()
        if small_inputs:
            feats['conv0'] = nn.Conv2d(3, num_init_features, kernel_size=3, stride=1, padding=1, bias=False)
        else:
            feats['conv0'] = nn.Conv2d(3, num_init_features, kernel_size=7, stride=2, padding=3, bias=False)
            feats['norm0'] = nn.BatchNorm2d(num_init_features)
            feats['relu0'] = nn.ReLU(inplace=True)
            feats['pool0'] = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=False)
        c = num_init_features
        self._blocks = []           # [(c_in, [layers], transition|None, c_total)]
        for bi, n_layers in enumerate(self.block_config):
            block = nn.Module()
            layers = []
            for li in range(n_layers):
                layer = _DenseLayer(c + li * growth_rate, growth_rate, bn_size, drop_rate)
                block.add_module('denselayer%d' % (li + 1), layer)
                layers.append(layer)
            feats['denseblock%d' % (bi + 1)] = block
            c_total = c + n_layers * growth_rate
            trans = None
            if bi != len(self.block_config) - 1:
                trans = _Transition(c_total, int(c_total * compression))
                feats['transition%d' % (bi + 1)] = trans
            self._blocks.append((c, layers, trans, c_total))
            c = int(c_total * compression) if trans is not None else c_total
        feats['norm_final'] = nn.BatchNorm2d(c)
        self.features = nn.Sequential(feats)
        self.num_features = c
        self.classifier = nn.Linear(c, num_classes)

        for name, param in self.named_parameters():          # densenet.py:141-150
            if 'conv' in name and 'weight' in name:
                n = param.size(0) * param.size(2) * param.size(3)
                param.data.normal_().mul_(math.sqrt(2. / n))
            elif 'norm' in name and 'weight' in name:
                param.data.fill_(1)
            elif 'norm' in name and 'bias' in name:
                param.data.fill_(0)
            elif 'classifier' in name and 'bias' in name:
                param.data.fill_(0)
        self._dropout_mask = None   # test hook: callable (layer index, rows, columns, device) -> bool keep-mask [rows][columns]
        self._cache = {}
        self._cache_epoch = 0       # part of every cache key; bumped by invalidate_cache()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_cache())

    # ------------------------------------------------------------------ cached, derived device tensors
    def invalidate_cache(self):
        """Drop the derived device tensors of the eval forward (folded BN scale/shift, repacked / Winograd / fp16 conv
        weights).  They are keyed on each source tensor's (`_version`, `data_ptr()`), which catches optimizer steps and
        `.to()`; writes that bump neither - kernels updating running statistics through raw pointers, `p.data` writes,
        collectives into `.data` - must call this.  Called by the training forward, `load_state_dict`, `_apply`
        (`.to()`, `.float()`, ...) and `distributed.broadcast_module`."""
        self._cache = {}
        self._cache_epoch += 1

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_cache()
        return out

    def _key(self, tensors):
        return (self._cache_epoch,) + tuple(v for t in tensors for v in (t._version, t.data_ptr()))

    def _bn_modules(self):
        mods = []
        if not self.small_inputs:
            mods.append(self.features.norm0)
        for _, layers, trans, _ in self._blocks:
            for l in layers:
                mods += [l.norm1, l.norm2]
            if trans is not None:
                mods.append(trans.norm)
        mods.append(self.features.norm_final)
        return mods

    def _folded_eval(self):
        """{bn module: (scale, shift)} for running-stat BN; refreshed when any BN tensor changed."""
        mods = self._bn_modules()
        key = self._key([t for m in mods for t in (m.weight, m.bias, m.running_mean, m.running_var)])
        hit = self._cache.get('fold')
        if hit is not None and hit[0] == key:
            return hit[1]
        total = sum((m.num_features + 3) // 4 * 4 for m in mods)   # every slice starts 16-B aligned (float4 loads)
        dev = mods[0].weight.device
        buf = torch.empty((2, total), device=dev, dtype=F32)
        table, off = {}, 0
        st = L.stream()
        for m in mods:
            c = m.num_features
            sc, sh = buf[0, off:off + c], buf[1, off:off + c]
            L.call('gnx_bn_fold_eval', c, L.ptr(m.weight), L.ptr(m.bias), L.ptr(m.running_mean),
                   L.ptr(m.running_var), float(m.eps), L.ptr(sc), L.ptr(sh), None, None, st)
            table[m] = (sc, sh)
            off += (c + 3) // 4 * 4
        self._cache['fold'] = (key, table, buf)
        return table

    def _repacked_conv2(self):
        """{layer: conv2 weight as [tap][growth][mid]} refreshed when a conv2 weight changed."""
        layers = [l for _, ls, _, _ in self._blocks for l in ls]
        key = self._key([l.conv2.weight for l in layers])
        hit = self._cache.get('w2')
        if hit is not None and hit[0] == key:
            return hit[1]
        table = {}
        st = L.stream()
        for l in layers:
            w = l.conv2.weight
            n, k = w.shape[0], w.shape[1]
            wr = torch.empty((9, n, k), device=w.device, dtype=F32)
            L.call('gnx_repack_conv3x3', L.ptr(w.detach().contiguous()), L.ptr(wr), n, k, st)
            table[l] = wr
        self._cache['w2'] = (key, table)
        return table

    def _repacked_conv2_f16(self):
        """{layer: the tap-major conv2 weight rounded to fp16} (config 5's DMA conv2), refreshed with the weights."""
        w2 = self._repacked_conv2()
        hit = self._cache.get('w2h')
        if hit is not None and hit[0] is w2:
            return hit[1]
        table = {l: w.to(torch.float16) for l, w in w2.items()}
        self._cache['w2h'] = (w2, table)
        return table

    def _conv1_f16(self):
        """{layer: conv1 weight [mid][cin] rounded to fp16} (config 5 on fp16 block buffers), refreshed with the weights."""
        layers = [l for _, ls, _, _ in self._blocks for l in ls]
        key = self._key([l.conv1.weight for l in layers])
        hit = self._cache.get('w1h')
        if hit is not None and hit[0] == key:
            return hit[1]
        table = {l: l.conv1.weight.detach().reshape(l.conv1.weight.shape[0], -1).to(torch.float16).contiguous()
                 for l in layers}
        self._cache['w1h'] = (key, table)
        return table

    def _dense_f16_packed(self):
        """{layer: (w1p, w2p)}: conv1 / conv2 weights rounded to fp16 once, in the MFMA-fragment order gnx_dense_layer_f16
        streams (csrc/dense_layer_f16.hip); refreshed with the weights."""
        layers = [l for _, ls, _, _ in self._blocks for l in ls]
        key = self._key([w for l in layers for w in (l.conv1.weight, l.conv2.weight)])
        hit = self._cache.get('dlp')
        if hit is not None and hit[0] == key:
            return hit[1]
        table = {}
        st = L.stream()
        for l in layers:
            w1, w2 = l.conv1.weight.detach().contiguous(), l.conv2.weight.detach().contiguous()
            k = w1.shape[1]
            w1p = torch.empty(w1.shape[0] * k, device=w1.device, dtype=torch.float16)
            w2p = torch.empty(w2.numel(), device=w1.device, dtype=torch.float16)
            L.call('gnx_dense_layer_f16_pack', L.ptr(w1), L.ptr(w2), L.ptr(w1p, torch.float16), L.ptr(w2p, torch.float16), k, st)
            table[l] = (w1p, w2p)
        self._cache['dlp'] = (key, table)
        return table

    def _trans_f16(self):
        """{transition: conv weight [c_out][c_in] rounded to fp16} (config 5's two-step transitions), refreshed with the weights."""
        trs = [t for _, _, t, _ in self._blocks if t is not None]
        key = self._key([t.conv.weight for t in trs])
        hit = self._cache.get('wth')
        if hit is not None and hit[0] == key:
            return hit[1]
        table = {t: t.conv.weight.detach().reshape(t.conv.weight.shape[0], -1).to(torch.float16).contiguous() for t in trs}
        self._cache['wth'] = (key, table)
        return table

    def _trans_f16_packed(self):
        """{transition: conv weight in gnx_transition_f16's fragment order (fp16)} - the fused transitions of config 5 -,
        refreshed with the weights."""
        trs = [t for _, _, t, _ in self._blocks if t is not None]
        key = self._key([t.conv.weight for t in trs])
        hit = self._cache.get('wtp')
        if hit is not None and hit[0] == key:
            return hit[1]
        st = L.stream()
        table = {}
        for t in trs:
            w = t.conv.weight.detach().reshape(t.conv.weight.shape[0], -1).contiguous()
            wp = torch.empty(w.numel(), device=w.device, dtype=torch.float16)
            L.call('gnx_transition_f16_pack', L.ptr(w), L.ptr(wp, torch.float16), w.shape[0], w.shape[1], st)
            table[t] = wp
        self._cache['wtp'] = (key, table)
        return table

    def _f16_dma_ok(self, M, s, mid, c_total):
        """Shapes gnx_conv3x3_f16_dma takes (conv3x3.hip): growth 32, 128 | mid, power-of-two maps 4..64, whole 128-row
        tiles, 32-bit element offsets."""
        return (self.growth_rate == 32 and mid % 128 == 0 and s in (4, 8, 16, 32, 64) and M % 128 == 0
                and M * max(mid, c_total) < 2 ** 31)

    def _winograd_conv2(self):
        """{layer: conv2 weight as Winograd F(2,3)-along-x factors [3][4][growth][mid]} refreshed with the weights."""
        layers = [l for _, ls, _, _ in self._blocks for l in ls]
        key = self._key([l.conv2.weight for l in layers])
        hit = self._cache.get('w2u')
        if hit is not None and hit[0] == key:
            return hit[1]
        table = {}
        st = L.stream()
        for l in layers:
            w = l.conv2.weight
            n, k = w.shape[0], w.shape[1]
            wu = torch.empty((12, n, k), device=w.device, dtype=F32)
            L.call('gnx_winograd_conv3x3_weights', L.ptr(w.detach().contiguous()), L.ptr(wu), n, k, st)
            table[l] = wu
        self._cache['w2u'] = (key, table)
        return table

    def _split_conv1(self):
        """{layer: conv1 weight split into bf16 hi / lo planes by 64-wide K chunks (gnx_conv1x1_split_pack)} refreshed with the
        weights."""
        layers = [l for _, ls, _, _ in self._blocks for l in ls]
        key = self._key([l.conv1.weight for l in layers])
        hit = self._cache.get('w1s')
        if hit is not None and hit[0] == key:
            return hit[1]
        table = {}
        st = L.stream()
        for l in layers:
            w = l.conv1.weight
            k = w.shape[1]
            wp = torch.empty(L.query('gnx_conv1x1_split_pack_halves', k), device=w.device, dtype=torch.bfloat16)
            L.call('gnx_conv1x1_split_pack', L.ptr(w.detach().contiguous()), wp.data_ptr(), k, st)
            table[l] = wp
        self._cache['w1s'] = (key, table)
        return table

    def _split_conv2(self):
        """{layer: conv2 weight split into bf16 hi / lo planes by 32-channel chunks and taps (gnx_conv3x3_split_pack)}."""
        layers = [l for _, ls, _, _ in self._blocks for l in ls]
        key = self._key([l.conv2.weight for l in layers])
        hit = self._cache.get('w2s')
        if hit is not None and hit[0] == key:
            return hit[1]
        table = {}
        st = L.stream()
        for l in layers:
            w = l.conv2.weight
            wp = torch.empty(L.query('gnx_conv3x3_split_pack_halves'), device=w.device, dtype=torch.bfloat16)
            L.call('gnx_conv3x3_split_pack', L.ptr(w.detach().contiguous()), wp.data_ptr(), st)
            table[l] = wp
        self._cache['w2s'] = (key, table)
        return table

    def _norm_vector(self, dev):
        """Device floats {mean[3], std[3], 1/std[3]} for the uint8 entry points, or None (ToTensor only)."""
        if self.input_norm is None:
            return None
        mean, std = self.input_norm
        key = (tuple(float(v) for v in mean), tuple(float(v) for v in std), str(dev), self._cache_epoch)
        hit = self._cache.get('nrm')
        if hit is not None and hit[0] == key:
            return hit[1]
        m = torch.tensor(key[0], dtype=F32)
        sd = torch.tensor(key[1], dtype=F32)
        assert m.numel() == 3 and sd.numel() == 3, "input_norm = (mean[3], std[3])"
        v = torch.cat([m, sd, 1.0 / sd]).to(dev)              # 1 / std: one correctly rounded fp32 division
        self._cache['nrm'] = (key, v)
        return v

    def _float_patches(self, x):
        """ToTensor (+ Normalize) of uint8 patches (N, 3, P, P) as its own pass -> float32, the floats torch would produce
        (gnx_u8_to_f32); float input passes through."""
        if x.dtype != torch.uint8:
            return x.contiguous().float()
        x = x.contiguous()
        out = torch.empty(x.shape, device=x.device, dtype=F32)
        if x.numel():
            L.call('gnx_u8_to_f32', x.data_ptr(), L.ptr(out), x.shape[0], x.shape[1], x.shape[2], x.shape[3],
                   L.ptr(self._norm_vector(x.device)), L.stream())
        return out

    def _geometry(self, P):
        if self.small_inputs:
            hs, s = None, P
        else:
            hs = (P + 6 - 7) // 2 + 1
            s = (hs + 2 - 3) // 2 + 1
        sizes = []
        for _ in self._blocks:
            sizes.append(s)
            s = s // 2
        return hs, sizes

    def _auto_chunk(self, P, n, elem_bytes=4):
        if self.atonce is not None:
            return max(1, min(int(self.atonce), n))
        hs, sizes = self._geometry(P)
        per_spot = 0
        if hs is not None:
            per_spot += hs * hs * self.features.conv0.out_channels
        for (c_in, layers, trans, c_total), s in zip(self._blocks, sizes):
            per_spot += s * s * c_total
        per_spot += sizes[0] * sizes[0] * self.bn_size * self.growth_rate
        # Later blocks have few positions per spot (S=4: 16), so a launch only fills 256 CUs when thousands of
        # spots go through together: size chunks by HBM (288 GB), not by cache - a whole 128-px array is 17 GB.
        budget = 40 * 1024 ** 3 // elem_bytes      # elements
        return max(1, min(n, max(32, budget // max(per_spot, 1))))

    # ------------------------------------------------------------------ optional per-kernel timing (bench.py)
    _probe = None      # when a list: (kind, start_event, end_event, flops, bytes) per dense-layer launch, on the launch stream
                       # (flops / bytes: the ALGORITHMIC work of that launch - bench.py credits a kernel kind with exactly the
                       # launches it timed)

    def _probe_begin(self):
        if self._probe is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _probe_mark(self, kind, start, flops=None, nbytes=None):
        if self._probe is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._probe.append((kind, start, ev, flops, nbytes))
        return ev

    # ------------------------------------------------------------------ stem of the eval forward
    def _stem_eval(self, xu, rows, c_total, P, hs, use_h, fold, w0, stem_out, chunk, st):
        """conv0 (-> norm0 -> relu0 -> pool0) of `xu` (float or uint8 patches) into the first block buffer `rows`.
        Returns the conv0-map scratch buffer (allocated on first need by the two-kernel path)."""
        nu = xu.shape[0]
        dev = xu.device
        c0 = self.features.conv0.out_channels
        if self.small_inputs:
            xu = self._float_patches(xu)
            L.call('gnx_conv_stem', L.ptr(xu), L.ptr(w0), L.ptr(rows), c_total, nu, 3, P, P, c0, 3, 3, 1, 1, st)
            return stem_out
        sc, sh = fold[self.features.norm0]
        if use_h and self.f16_stem:
            # config 5: the stem's matrix operands in fp16 too (float or uint8 patches)
            u8 = xu.dtype == torch.uint8
            rc = L.query('gnx_conv_stem_bnrelu_maxpool_f16mul', xu.data_ptr(), 1 if u8 else 0, L.ptr(w0), rows.data_ptr(),
                         c_total, nu, 3, P, P, c0, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh),
                         L.ptr(self._norm_vector(dev)) if u8 else None, st)
            if rc == 0:
                return stem_out
            if rc != L.ERR_UNSUPPORTED:
                raise RuntimeError("gnx_conv_stem_bnrelu_maxpool_f16mul failed (%d)" % rc)
        # conv0 -> norm0 -> relu0 -> pool0 in one kernel where the geometry allows (128- / 256-px patches): the conv0 map
        # (5.2 GB per 128-px array) then never touches HBM.  uint8 patches: ToTensor (+ Normalize) inside that kernel too.
        if xu.dtype == torch.uint8:
            rc = L.query('gnx_conv_stem_bnrelu_maxpool_u8', xu.data_ptr(), L.ptr(w0), rows.data_ptr(), c_total, nu, 3, P, P,
                         c0, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh), L.ptr(self._norm_vector(dev)), 1 if use_h else 0, st)
            if rc == 0:
                return stem_out
            if rc != L.ERR_UNSUPPORTED:
                raise RuntimeError("gnx_conv_stem_bnrelu_maxpool_u8 failed (%d)" % rc)
            xu = self._float_patches(xu)                      # other geometries: convert, then the float stems
        if use_h:
            L.call('gnx_conv_stem_bnrelu_maxpool_h16', L.ptr(xu), L.ptr(w0), L.ptr(rows, torch.float16), c_total, nu, 3, P, P,
                   c0, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh), st)
            return stem_out
        rc = L.query('gnx_conv_stem_bnrelu_maxpool', L.ptr(xu), L.ptr(w0), L.ptr(rows), c_total, nu, 3, P, P, c0, 7, 7, 2, 3,
                     L.ptr(sc), L.ptr(sh), st)
        if rc == L.ERR_UNSUPPORTED:
            if stem_out is None:
                stem_out = torch.empty((chunk * hs * hs, c0), device=dev, dtype=F32)
            L.call('gnx_conv_stem', L.ptr(xu), L.ptr(w0), L.ptr(stem_out), c0, nu, 3, P, P, c0, 7, 7, 2, 3, st)
            L.call('gnx_bnrelu_maxpool', L.ptr(stem_out), c0, L.ptr(rows), c_total, nu, c0, hs, hs, L.ptr(sc), L.ptr(sh), st)
        elif rc != 0:
            raise RuntimeError("gnx_conv_stem_bnrelu_maxpool failed (%d)" % rc)
        return stem_out

    def _block_fused(self, bi, buf, nxt, xs, n, s, P, fold, w0, dlp, st):
        """One dense block of config 5 on its channel-blocked fp16 buffer `buf` [c_total / 32][rows][32] for the chunk's `n`
        spots: the fp16 stem (block 0) or nothing (the previous transition already stored this block's first channels), every
        dense layer as one kernel (gnx_dense_layer_f16), and the transition into `nxt` in two steps (pooling pass reading
        the blocked buffer, 1x1 conv storing blocked)."""
        c_in, layers, trans, c_total = self._blocks[bi]
        H = torch.float16
        rows_total = buf.shape[1]
        if bi == 0:
            sc, sh = fold[self.features.norm0]
            u8 = xs.dtype == torch.uint8
            c0 = self.features.conv0.out_channels
            L.call('gnx_conv_stem_bnrelu_maxpool_f16mul_cb', xs.data_ptr(), 1 if u8 else 0, L.ptr(w0), buf.data_ptr(), rows_total, n,
                   3, P, P, c0, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh), L.ptr(self._norm_vector(xs.device)) if u8 else None, st)
        for li, layer in enumerate(layers):
            cin = c_in + li * self.growth_rate
            sc1, sh1 = fold[layer.norm1]
            sc2, sh2 = fold[layer.norm2]
            t0 = self._probe_begin()
            L.call('gnx_dense_layer_f16', L.ptr(buf, H), rows_total, n, s, cin, L.ptr(dlp[layer][0], H), L.ptr(dlp[layer][1], H),
                   L.ptr(sc1), L.ptr(sh1), L.ptr(sc2), L.ptr(sh2), st)
            Ml = n * s * s
            self._probe_mark('dense_layer', t0, 2 * Ml * (cin * 128 + 9 * 128 * 32), 2 * Ml * (cin + 32))
        if trans is not None:
            so = s // 2
            sct, sht = fold[trans.norm]
            cout = trans.conv.out_channels
            if (self.f16_fused_transitions and s in (8, 16, 32, 64) and 64 <= c_total <= 1024 and cout % 128 == 0 and
                    cout <= 512 and (n * so * so) % 128 == 0 and rows_total * 64 < 2 ** 32 - 2 ** 25):
                # one kernel: the pooled activated operand exists in the LDS only (bit-identical to the pooling pass's output)
                L.call('gnx_transition_f16', L.ptr(buf, H), rows_total, n, s, c_total, cout, L.ptr(self._trans_f16_packed()[trans], H),
                       L.ptr(sct), L.ptr(sht), L.ptr(nxt, H), nxt.shape[1], st)
                return
            pooled = torch.empty((n * so * so, c_total), device=buf.device, dtype=H)
            L.call('gnx_bnrelu_avgpool2_h16_cb', L.ptr(buf, H), rows_total, L.ptr(pooled, H), c_total, n, c_total, s, L.ptr(sct),
                   L.ptr(sht), st)
            L.call('gnx_conv1x1_bnrelu_h16_cb', L.ptr(pooled, H), c_total, L.ptr(self._trans_f16()[trans], H), L.ptr(nxt, H),
                   nxt.shape[1], n * so * so, cout, c_total, None, None, None, None, st)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("gridnext_amd.DenseNet runs on a HIP device only (input is on %s); "
                               "there is no CPU fallback" % x.device)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3]:
            raise ValueError("expected square RGB patches (N, 3, P, P), got %s" % (tuple(x.shape),))
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.float()
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        # fp16 path, running statistics: its kernels take whole groups of 8 spots (128-row tiles on the 4 x 4 maps).  A ragged
        # batch is padded with empty patches and the extra rows dropped: spots are independent under running statistics, and on
        # the gradient path the padding rows receive a zero output gradient, so they add nothing to any parameter gradient.
        pad = (-x.shape[0]) % 8
        if self.mfma == 'f16' and pad and x.shape[0] > 0 and not self.training and not x.requires_grad and self.f16_buffers \
                and not self.small_inputs and x.shape[2] in (128, 256) and self.growth_rate == 32 \
                and (self.bn_size * self.growth_rate) % 128 == 0:     # (only calls the fp16-buffer kernels can take: others
            #                                                            would pay the copy and 7 extra spots for nothing)
            xp = torch.cat([x, x.new_zeros((pad,) + tuple(x.shape[1:]))], 0)
            return self.forward(xp)[:x.shape[0]]
        if self.training or needs_grad:
            from .densenet_train import densenet_autograd       # training / gradient path
            return densenet_autograd(self, x)
        return self._forward_eval(x.detach())

    @torch.no_grad()
    def _forward_eval(self, x):
        # uint8 patches stay uint8 up to the stem kernel's operand load (a quarter of the bytes over PCIe and out of HBM)
        x = x.contiguous() if x.dtype == torch.uint8 else x.contiguous().float()
        N, _, P, _ = x.shape
        dev = x.device
        st = L.stream()
        fold = self._folded_eval()
        w2 = self._repacked_conv2()
        w2u = self._winograd_conv2() if (self.winograd and self.mfma == 'f32') else None
        w2h = self._repacked_conv2_f16() if self.mfma == 'f16' else None
        w1s = self._split_conv1() if (self.split_conv1 and self.mfma == 'f32' and self.bn_size * self.growth_rate == 128) else None
        w2s = self._split_conv2() if (self.split_conv2 and self.mfma == 'f32' and self.bn_size * self.growth_rate == 128 and
                                      self.growth_rate == 32) else None
        if self.mfma not in ('f32', 'f16'):
            raise ValueError("DenseNet.mfma must be 'f32' or 'f16'")
        sfx = '_f16' if self.mfma == 'f16' else ''
        hs, sizes = self._geometry(P)
        mid = self.bn_size * self.growth_rate
        conv0 = self.features.conv0
        c0 = conv0.out_channels
        # the fused dense-layer kernel (fp16 block buffers) takes: growth 32, bottleneck 128, maps of 4..64, 32 | channels >= 64;
        # its stem (gnx_conv_stem_bnrelu_maxpool_f16mul_cb) 32 | c0 <= 64 and always multiplies fp16 operands (f16_stem)
        fused_ok = bool(sfx) and self.f16_buffers and self.f16_fused and self.growth_rate == 32 and mid == 128 and \
            all(s in (4, 8, 16, 32, 64) for s in sizes) and all(blk[0] % 32 == 0 and blk[0] >= 64 for blk in self._blocks) and \
            self.num_features % 32 == 0 and not self.small_inputs and P in (128, 256) and \
            all(blk[3] <= 1024 + 32 for blk in self._blocks) and c0 in (32, 64) and self.f16_stem
        # fp16 BLOCK BUFFERS (decided for the whole call: the buffers cannot change type half-way) hold twice the spots in
        # the same bytes - a whole 256-px array, 34 GB, is then one chunk; the chunk is sized for the element type that is
        # actually taken
        h_shapes = bool(sfx) and self.f16_buffers and not self.small_inputs and P in (128, 256) and c0 % 4 == 0 and \
            self.growth_rate == 32 and mid % 128 == 0 and N % 8 == 0 and all(s in (4, 8, 16, 32, 64) for s in sizes)
        chunk = self._auto_chunk(P, N, 2 if h_shapes else 4)
        if not sfx and self.atonce is None and 128 <= chunk < N:
            # fp32 path, a batch that goes through in chunks anyway: chunks of whole groups of 128 spots - every map then has
            # whole 128-row tiles even for 7 x 7 maps (224-px patches: the reference's own geometry) - whose element offsets fit
            # the 32 bits the LDS-DMA conv2 indexes with.  Otherwise every launch falls back to the generic kernels
            # (conv1x1_kernel / conv3x3_pipe_kernel: 0.69 / 0.72 of the matrix peak against 0.78 / 0.89).
            lim = min((2 ** 31 - 1) // (sz * sz * max(mid, blk[3])) for blk, sz in zip(self._blocks, sizes))
            chunk = max(128, min(chunk, lim) // 128 * 128)
        if sfx and self.atonce is None and chunk >= 8:
            if fused_ok and h_shapes:
                chunk = chunk // 8 * 8                          # whole 128-row tiles; it indexes with 64 bits
            else:
                # two-kernel fp16 path: chunks of whole 128-row tiles whose element offsets fit 32 bits (its DMA kernels)
                lim = min((2 ** 31 - 1) // (sz * sz * max(mid, blk[3])) for blk, sz in zip(self._blocks, sizes))
                chunk = max(8, min(chunk, lim) // 8 * 8)
        conv0 = self.features.conv0
        c0 = conv0.out_channels
        def sub_range(bi, n):
            return n                                            # (a block's whole layer chain over all the chunk's spots)

        # config 5 with fp16 BLOCK BUFFERS: the concatenated features live in HBM as fp16 (as under the reference's autocast),
        # every kernel of the chain reads / writes halves.  Taken when every launch of the call has a shape those kernels
        # take (decided here, for the whole call: the buffers cannot change type half-way).
        use_h = h_shapes and chunk % 8 == 0 and \
            (fused_ok or all(sub_range(bi, min(chunk, N)) * sizes[bi] ** 2 * max(mid, self._blocks[bi][3]) < 2 ** 31
                             for bi in range(len(sizes))))
        fused = use_h and fused_ok
        self._used_f16_buffers = use_h                          # introspection (tests, bench)
        self._used_f16_fused = fused
        w1h = self._conv1_f16() if (use_h and not fused) else None
        dlp = self._dense_f16_packed() if fused else None
        # workspace for one chunk
        if fused:
            # channel-blocked block buffers [c_total / 32][rows][32] (include/gridnext_hip.h: gnx_dense_layer_f16): the 32
            # channels a layer's K-loop stage needs of consecutive pixels are contiguous memory
            bufs = [torch.empty((c_total // 32, chunk * s * s, 32), device=dev, dtype=torch.float16)
                    for (_, _, _, c_total), s in zip(self._blocks, sizes)]
        else:
            bufs = [torch.empty((chunk * s * s, c_total), device=dev, dtype=torch.float16 if use_h else F32)
                    for (_, _, _, c_total), s in zip(self._blocks, sizes)]
        bott = torch.empty((1 if fused else chunk * sizes[0] * sizes[0], mid), device=dev, dtype=F32)
        bott16 = bott.view(torch.float16)                       # the same memory as [rows][2 mid] halves (fp16 path)
        stem_out = None                                         # conv0 map: only the unfused stem needs it
        feats = torch.empty((N, self.num_features), device=dev, dtype=F32)
        w0 = conv0.weight.detach().contiguous()

        for s0 in range(0, N, chunk):
            n = min(chunk, N - s0)
            xs = x[s0:s0 + n]
            for bi, ((c_in, layers, trans, c_total), s) in enumerate(zip(self._blocks, sizes)):
                buf = bufs[bi]
                sub = sub_range(bi, n)
                for u0 in range(0, n, sub):
                    nu = min(sub, n - u0)
                    M = nu * s * s
                    if fused:
                        self._block_fused(bi, buf, bufs[bi + 1] if trans is not None else None, xs, nu, s, P, fold, w0, dlp, st)
                        continue
                    rows = buf[u0 * s * s:(u0 + nu) * s * s]
                    if bi == 0:
                        stem_out = self._stem_eval(xs[u0:u0 + nu], rows, c_total, P, hs, use_h, fold, w0, stem_out, chunk, st)
                    for li, layer in enumerate(layers):
                        cin = c_in + li * self.growth_rate
                        sc1, sh1 = fold[layer.norm1]
                        sc2, sh2 = fold[layer.norm2]
                        t0 = self._probe_begin()
                        eb = 2 if use_h else 4                                        # bytes per block-buffer element
                        em = 2 if (use_h or (sfx and self._f16_dma_ok(M, s, mid, c_total))) else 4   # ... per bottleneck element
                        work1 = (2 * M * cin * mid, M * (cin * eb + mid * em))
                        work2 = (2 * M * 9 * mid * self.growth_rate, M * (mid * em + self.growth_rate * eb))
                        if use_h:
                            L.call('gnx_conv1x1_bnrelu_h16', L.ptr(rows, torch.float16), c_total,
                                   L.ptr(w1h[layer], torch.float16), L.ptr(bott16, torch.float16), mid, M, mid, cin,
                                   L.ptr(sc1), L.ptr(sh1), L.ptr(sc2), L.ptr(sh2), st)
                            t1 = self._probe_mark('conv1x1', t0, *work1)
                            L.call('gnx_conv3x3_f16_dma_h', L.ptr(bott16, torch.float16), mid,
                                   L.ptr(w2h[layer], torch.float16), rows.data_ptr() + 2 * cin, c_total, M,
                                   self.growth_rate, mid, s, st)
                        elif sfx and self._f16_dma_ok(M, s, mid, c_total):
                            # fp16 bottleneck: conv1 stores it activated and rounded, conv2 streams it by DMA.  The choice
                            # depends on the map size and channel counts only (128 | M holds for every whole spot)
                            L.call('gnx_conv1x1_bnrelu_f16_act16', L.ptr(rows), c_total, L.ptr(layer.conv1.weight),
                                   L.ptr(bott16, torch.float16), mid, M, mid, cin, L.ptr(sc1), L.ptr(sh1), L.ptr(sc2), L.ptr(sh2), st)
                            t1 = self._probe_mark('conv1x1', t0, *work1)
                            L.call('gnx_conv3x3_f16_dma', L.ptr(bott16, torch.float16), mid, L.ptr(w2h[layer], torch.float16),
                                   rows.data_ptr() + 4 * cin, c_total, M, self.growth_rate, mid, s, st)
                        elif sfx:
                            L.call('gnx_conv1x1_bnrelu_f16', L.ptr(rows), c_total, L.ptr(layer.conv1.weight),
                                   L.ptr(bott), mid, M, mid, cin, L.ptr(sc1), L.ptr(sh1), 0, 0, st)
                            t1 = self._probe_mark('conv1x1', t0, *work1)
                            L.call('gnx_conv3x3_bnrelu_f16', L.ptr(bott), mid, L.ptr(w2[layer]),
                                   rows.data_ptr() + 4 * cin, c_total, M, self.growth_rate, mid, s, L.ptr(sc2),
                                   L.ptr(sh2), st)
                        else:
                            # norm2 + relu2 ride on conv1's store: conv2 then takes its operand as it lies in HBM
                            # (global -> LDS DMA, no prologue)
                            rc = L.ERR_UNSUPPORTED
                            if w1s is not None:
                                # (opt-in) the same product on split bf16 operands; shapes it declines keep the fp32 instruction
                                rc = L.query('gnx_conv1x1_bnrelu_act_split', L.ptr(rows), c_total, w1s[layer].data_ptr(), L.ptr(bott),
                                             mid, M, cin, L.ptr(sc1), L.ptr(sh1), L.ptr(sc2), L.ptr(sh2), st)
                                if rc not in (0, L.ERR_UNSUPPORTED):
                                    raise RuntimeError("gnx_conv1x1_bnrelu_act_split failed (%d)" % rc)
                            if rc == L.ERR_UNSUPPORTED:
                                L.call('gnx_conv1x1_bnrelu_act', L.ptr(rows), c_total, L.ptr(layer.conv1.weight),
                                       L.ptr(bott), mid, M, mid, cin, L.ptr(sc1), L.ptr(sh1), L.ptr(sc2), L.ptr(sh2), st)
                            t1 = self._probe_mark('conv1x1', t0, *work1)
                            # conv2 on the ready operand: Winograd F(2,3) along x (1.5x fewer matrix operations, rounding-
                            # level differences) for maps of 8 x 8 and up (4 x 4 measured faster direct).  The choice
                            # depends on the map size only - never on how many spots a call or a chunk holds - so chunked
                            # and unchunked evaluation stay bit-identical.
                            rc = L.ERR_UNSUPPORTED
                            if w2s is not None:
                                # (opt-in) nine shifted products of split bf16 operands; declined shapes fall through
                                rc = L.query('gnx_conv3x3_split', L.ptr(bott), mid, w2s[layer].data_ptr(), rows.data_ptr() + 4 * cin,
                                             c_total, M, s, st)
                                if rc not in (0, L.ERR_UNSUPPORTED):
                                    raise RuntimeError("gnx_conv3x3_split failed (%d)" % rc)
                            if rc == L.ERR_UNSUPPORTED and w2u is not None and s >= 8:
                                rc = L.query('gnx_conv3x3_winograd', L.ptr(bott), mid, L.ptr(w2u[layer]),
                                             rows.data_ptr() + 4 * cin, c_total, M, self.growth_rate, mid, s, st)
                                if rc not in (0, L.ERR_UNSUPPORTED):
                                    raise RuntimeError("gnx_conv3x3_winograd failed (%d)" % rc)
                            if rc == L.ERR_UNSUPPORTED:
                                L.call('gnx_conv3x3_bnrelu', L.ptr(bott), mid, L.ptr(w2[layer]),
                                       rows.data_ptr() + 4 * cin, c_total, M, self.growth_rate, mid, s, None, None, st)
                        self._probe_mark('conv3x3', t1, *work2)
                    if trans is not None:
                        nxt = bufs[bi + 1]
                        so = s // 2
                        sct, sht = fold[trans.norm]
                        # transitions are HBM-bound (4x the input bytes of their output): the fp32 wave-specialised
                        # kernel serves both matrix precisions
                        if use_h and c_total % 32 == 0:
                            # two steps: norm -> relu -> 2x2 mean in one pass over the block buffer (16-B accesses), then the
                            # 1x1 conv on the pooled rows without prologue or consumer activation
                            pooled = torch.empty((nu * so * so, c_total), device=dev, dtype=torch.float16)
                            L.call('gnx_bnrelu_avgpool2_h16', L.ptr(rows, torch.float16), c_total, L.ptr(pooled, torch.float16),
                                   c_total, nu, c_total, s, L.ptr(sct), L.ptr(sht), st)
                            L.call('gnx_conv1x1_bnrelu_h16', L.ptr(pooled, torch.float16), c_total,
                                   L.ptr(self._trans_f16()[trans], torch.float16), L.ptr(nxt[u0 * so * so:], torch.float16),
                                   nxt.shape[1], nu * so * so, trans.conv.out_channels, c_total, None, None, None, None, st)
                            del pooled
                        elif use_h:
                            L.call('gnx_conv1x1_bnrelu_f16_h', L.ptr(rows, torch.float16), c_total,
                                   L.ptr(trans.conv.weight), L.ptr(nxt[u0 * so * so:], torch.float16), nxt.shape[1],
                                   nu * so * so, trans.conv.out_channels, c_total, L.ptr(sct), L.ptr(sht), None, None, 1, s,
                                   st)
                        else:
                            L.call('gnx_conv1x1_bnrelu', L.ptr(rows), c_total, L.ptr(trans.conv.weight),
                                   L.ptr(nxt[u0 * so * so:]), nxt.shape[1], nu * so * so, trans.conv.out_channels,
                                   c_total, L.ptr(sct), L.ptr(sht), 1, s, st)
            scf, shf = fold[self.features.norm_final]
            s_last = sizes[-1]
            if fused:
                L.call('gnx_bnrelu_avgpool_h16_cb', L.ptr(bufs[-1], torch.float16), bufs[-1].shape[1], L.ptr(feats[s0:]),
                       self.num_features, n, self.num_features, s_last * s_last, L.ptr(scf), L.ptr(shf), st)
            elif use_h:
                L.call('gnx_bnrelu_avgpool_h16', L.ptr(bufs[-1], torch.float16), bufs[-1].shape[1], L.ptr(feats[s0:]),
                       self.num_features, n, self.num_features, s_last * s_last, L.ptr(scf), L.ptr(shf), st)
            else:
                L.call('gnx_bnrelu_avgpool', L.ptr(bufs[-1]), bufs[-1].shape[1], L.ptr(feats[s0:]), self.num_features,
                       n, self.num_features, s_last * s_last, L.ptr(scf), L.ptr(shf), st)
        if not self.classify:
            return feats
        return GF.linear(feats, self.classifier.weight.detach(), self.classifier.bias.detach())
