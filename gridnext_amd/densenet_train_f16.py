"""DenseNet-BC gradient path on the fp16-MFMA kernels (BASELINE config 5 with f trained).

`DenseNet.mfma = 'f16'` on the gradient path under running statistics - `train_gridwise` with `f_opt`
(/root/reference/gridnext/training.py:126 keeps f in eval mode, :164-171 steps it) - runs here: the taped forward of
/root/reference/gridnext/densenet.py:35-54 IS the eval forward of config 5 - every dense layer ONE kernel on channel-blocked
fp16 block buffers [c_total / 32][rows][32] (gnx_dense_layer_f16_tape: the same kernel, bit for bit, which also copies the
activated bottleneck tile it holds in the LDS out as the tape, [4][rows][32] halves) - and the backward of
csrc/dense_bwd_f16.hip runs on the SAME buffers (`_lb` entry points: (ld, bs) addressing) - fp16 matrix operands and fp16
gradient tensors, fp32 accumulation, fp32 parameter gradients.  (Round 4 ran the unfused forward pair on row-major buffers:
conv1 wrote the bottleneck, conv2 read it back.)

Scaling policy for the fp16 gradients: ONE power-of-two loss scale `s` per backward, chosen on the device from the gradient
that enters the network (no host synchronisation): s = 2^floor(12 - log2(max |dfeats| * max |scale_final| / S^2)), i.e. the
largest element of the last block's gradient lands in [2^11, 2^12) - a factor 16 of headroom before fp16 overflows and 26
binades down to the smallest normal.  Every fp16 gradient tensor holds s x the true gradient; every fp32 result is multiplied
by 1/s (exact) where its partial sums are reduced.  `model.f16_grad_overflow` (device int32) is OR-ed with 1 by any kernel
that reduces a non-finite value - STICKY across backwards (gradient accumulation), read and cleared by the training loop
before every optimizer step (`training._F16StepGuard`: an overflowed step is skipped and `model.f16_grad_target`, the
exponent 12 above, is lowered by one); `model.f16_grad_scale` holds {s, 1/s} of the last backward.  At every transition the scale
is re-centred for the block in front (another power of two, from the largest element of the pooled gradient that enters it):
`model.f16_grad_block_scales` lists the blocks' {s, 1/s}, last block first.

The stem (conv0 .. pool0, 64 channels) runs on fp16 matrix operands too: the forward is the inference kernel
(gnx_conv_stem_bnrelu_maxpool_f16mul) writing block 1's first columns, the backward (gnx_stem_bwd_f16) one pass over the
patches that recomputes the conv0 rows it needs, finds pool0's winners and contracts the routed gradient with the im2col of
the staged rows - no window indices, no fp32 conv0-map gradient in HBM.  Other stem widths (and `model.f16_stem = False`) keep
the fp32 stem with recorded window indices and its fp32 adjoints (gnx_h16_cols_to_f32 hands them the gradient).  The
classifier is fp32.
"""
import torch
from torch.autograd import Function

from . import _lib as L
from .densenet_train import _Tape, _bn, gammas_nonzero

F32, H16 = torch.float32, torch.float16


def eligible(model, x):
    """Shapes / modes the fp16 gradient path takes; anything else runs the fp32 path (densenet_train._DenseNetFn).  The taped
    forward is the fused dense-layer kernel on channel-blocked buffers (gnx_dense_layer_f16_tape): its limits apply."""
    if model.mfma != 'f16' or model.training or model.small_inputs or x.requires_grad:
        return False
    if model.growth_rate != 32 or model.bn_size * model.growth_rate != 128 or model.drop_rate > 0:
        return False
    N, _, P, _ = x.shape
    if P not in (128, 256) or N % 8 != 0 or N == 0:
        return False
    c0 = model.features.conv0.out_channels
    if c0 % 32 != 0 or c0 < 64 or model.num_features % 32 != 0:
        return False
    if any(blk[0] % 32 != 0 or blk[0] < 64 or blk[3] > 1024 + 32 for blk in model._blocks):
        return False
    hs, sizes = model._geometry(P)
    if any(s not in (4, 8, 16, 32, 64) for s in sizes):
        return False
    if any(t is not None and (t.conv.out_channels % 32 != 0) for _, _, t, _ in model._blocks):
        return False
    return gammas_nonzero(model)


def tape_bytes_per_spot(model, P):
    """HBM one spot holds on the fp16 tape plus its share of the backward's scratch: block buffers and their gradients (2 B),
    one activated bottleneck per layer (2 B), the pooled transition operands, the float patch and the fp32 pooled stem map."""
    hs, sizes = model._geometry(P)
    total, biggest = 0, 0
    for (c_in, layers, trans, c_total), s in zip(model._blocks, sizes):
        block = s * s * c_total
        total += block + len(layers) * s * s * 128 + (block // 4 if trans is not None else 0)
        biggest = max(biggest, 2 * block + 2 * s * s * 128)
    c0 = model.features.conv0.out_channels
    return 2 * (total + biggest) + 4 * 3 * P * P + 4 * 3 * sizes[0] * sizes[0] * c0


def _f32(n, dev):
    return torch.empty(max(int(n), 1), device=dev, dtype=F32)


def _rows_of_blocks(t, nblocks):
    """The first `nblocks` channel blocks of a channel-blocked buffer [C / 32][rows][32] as a row-major [rows][32 nblocks]
    matrix (one copy; for the few narrow operands whose consumers read rows: a transition's output gradient, the stem's)."""
    return t[:nblocks].permute(1, 0, 2).reshape(t.shape[1], 32 * nblocks).contiguous()


class _DenseNetF16Fn(Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        x = model._float_patches(x)
        N, _, P, _ = x.shape
        dev = x.device
        st = L.stream()
        hs, sizes = model._geometry(P)
        g, mid = model.growth_rate, model.bn_size * model.growth_rate
        conv0 = model.features.conv0
        c0 = conv0.out_channels
        tape = _Tape()
        tape.x, tape.N, tape.P, tape.hs, tape.sizes = x, N, P, hs, sizes
        # channel-blocked block buffers [c_total / 32][rows][32] (include/gridnext_hip.h: gnx_dense_layer_f16): the buffers the
        # eval forward of config 5 runs on - the taped forward IS that forward (the same kernels, bit for bit), plus the tape
        bufs = [torch.empty((c_total // 32, N * s * s, 32), device=dev, dtype=H16)
                for (_, _, _, c_total), s in zip(model._blocks, sizes)]
        tape.bufs = bufs
        w0 = conv0.weight.detach().contiguous()
        hp = (hs + 2 - 3) // 2 + 1
        s0 = _bn(model.features.norm0, None, c0, N * hs * hs, False, dev, st)
        tape.stats0 = s0
        tape.stem32 = tape.pool_idx = None
        if c0 == 64 and model.f16_stem:
            # ---- stem on fp16 matrix operands, straight into block 1's first channel blocks; its backward (gnx_stem_bwd_f16)
            #      recomputes the conv0 rows it needs from the patches - no window indices, no fp32 pooled map
            L.call('gnx_conv_stem_bnrelu_maxpool_f16mul_cb', L.ptr(x), 0, L.ptr(w0), bufs[0].data_ptr(), bufs[0].shape[1], N, 3, P, P,
                   c0, 7, 7, 2, 3, L.ptr(s0[0]), L.ptr(s0[1]), None, st)
        else:
            # ---- stem in fp32 with window indices (running statistics), then into block 1's first channel blocks as fp16
            stem32 = torch.empty((N * hp * hp, c0), device=dev, dtype=F32)
            tape.pool_idx = torch.empty((N * hp * hp, c0), device=dev, dtype=torch.uint8)
            L.call('gnx_conv_stem_bnrelu_maxpool_argmax', L.ptr(x), L.ptr(w0), L.ptr(stem32), c0, tape.pool_idx.data_ptr(), N, 3, P,
                   P, c0, 7, 7, 2, 3, L.ptr(s0[0]), L.ptr(s0[1]), st)
            bufs[0][:c0 // 32].copy_(stem32.view(N * hp * hp, c0 // 32, 32).permute(1, 0, 2))
            tape.stem32 = stem32
        dlp = model._dense_f16_packed()          # {layer: (conv1, conv2) weights in the fused kernel's fragment order, fp16}
        wth = model._trans_f16()                 # {transition: conv.weight [c_out][c_total] halves}
        tape.layers, tape.trans = [], []
        for bi, ((c_in, layers, trans, c_total), s) in enumerate(zip(model._blocks, sizes)):
            buf = bufs[bi]
            M = N * s * s
            recs = []
            for li, layer in enumerate(layers):
                cin = c_in + li * g
                s1 = _bn(layer.norm1, None, cin, M, False, dev, st)
                s2 = _bn(layer.norm2, None, mid, M, False, dev, st)
                # ONE kernel per dense layer (norm1 .. conv2, the bottleneck tile in the LDS); its tape: the activated
                # bottleneck, copied out of the LDS tile as [4][M][32] halves
                a = torch.empty((mid // 32, M, 32), device=dev, dtype=H16)
                t0 = model._probe_begin()
                L.call('gnx_dense_layer_f16_tape', buf.data_ptr(), M, N, s, cin, dlp[layer][0].data_ptr(), dlp[layer][1].data_ptr(),
                       L.ptr(s1[0]), L.ptr(s1[1]), L.ptr(s2[0]), L.ptr(s2[1]), a.data_ptr(), M, st)
                model._probe_mark('dense_layer_tape', t0, 2 * M * (cin * mid + 9 * mid * g), 2 * M * (cin + g + mid))
                recs.append((a, s1, s2))
            tape.layers.append(recs)
            if trans is not None:
                nxt = bufs[bi + 1]
                so = s // 2
                stt = _bn(trans.norm, None, c_total, M, False, dev, st)
                pooled = torch.empty((N * so * so, c_total), device=dev, dtype=H16)     # (row-major: the weight gradient's operand)
                cout = trans.conv.out_channels
                if (model.f16_fused_transitions and s in (8, 16, 32, 64) and 64 <= c_total <= 1024 and cout % 128 == 0 and
                        cout <= 512 and (N * so * so) % 128 == 0 and M * 64 < 2 ** 32 - 2 ** 25):
                    # ONE kernel (the eval forward's gnx_transition_f16), the pooled operand copied out of its LDS slots
                    L.call('gnx_transition_f16_tape', buf.data_ptr(), M, N, s, c_total, cout, model._trans_f16_packed()[trans].data_ptr(),
                           L.ptr(stt[0]), L.ptr(stt[1]), nxt.data_ptr(), nxt.shape[1], pooled.data_ptr(), c_total, st)
                else:
                    L.call('gnx_bnrelu_avgpool2_h16_cb', buf.data_ptr(), M, pooled.data_ptr(), c_total, N, c_total, s, L.ptr(stt[0]),
                           L.ptr(stt[1]), st)
                    L.call('gnx_conv1x1_bnrelu_h16_cb', pooled.data_ptr(), c_total, wth[trans].data_ptr(), nxt.data_ptr(), nxt.shape[1],
                           N * so * so, cout, c_total, None, None, None, None, st)
                tape.trans.append((stt, pooled))
            else:
                tape.trans.append(None)
        s_last, c_last = sizes[-1], model.num_features
        sf = _bn(model.features.norm_final, None, c_last, N * s_last * s_last, False, dev, st)
        tape.statsf = sf
        feats = torch.empty((N, c_last), device=dev, dtype=F32)
        L.call('gnx_bnrelu_avgpool_h16_cb', bufs[-1].data_ptr(), bufs[-1].shape[1], L.ptr(feats), c_last, N, c_last,
               s_last * s_last, L.ptr(sf[0]), L.ptr(sf[1]), st)
        tape.feats = feats
        tape.versions = [(p, p._version, p.data_ptr()) for p in params]
        ctx.tape, ctx.model = tape, model
        if not model.classify:
            return feats.clone()
        nc = model.classifier.out_features
        out = torch.empty((N, nc), device=dev, dtype=F32)
        L.call('gnx_gemm_f32', L.ptr(feats), c_last, 0, L.ptr(model.classifier.weight), c_last, 0, L.ptr(model.classifier.bias),
               L.ptr(out), nc, N, nc, c_last, 0, st)
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
        dout = dout.contiguous()
        dev = dout.device
        st = L.stream()
        N, P, hs, sizes = tape.N, tape.P, tape.hs, tape.sizes
        g, mid = model.growth_rate, model.bn_size * model.growth_rate
        grads = {}

        def want(p):
            return p is not None and p.requires_grad

        def new_like(p):
            t = torch.empty_like(p, memory_format=torch.contiguous_format)
            grads[p] = t
            return t

        def bn_out(bn):
            return (new_like(bn.weight) if want(bn.weight) else None, new_like(bn.bias) if want(bn.bias) else None)

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

        flag = model.__dict__.get('f16_grad_overflow')
        if flag is None or flag.device != dev:
            flag = model.__dict__['f16_grad_overflow'] = torch.zeros(1, device=dev, dtype=torch.int32)
        fp = flag.data_ptr()

        # ---- classifier (fp32)
        c_last = model.num_features
        if model.classify:
            cls = model.classifier
            nc = cls.out_features
            dfeats = torch.empty((N, c_last), device=dev, dtype=F32)
            L.call('gnx_gemm_f32', L.ptr(dout), nc, 0, L.ptr(cls.weight), c_last, 1, None, L.ptr(dfeats), c_last, N, c_last, nc, 0, st)
            if want(cls.weight):
                L.call('gnx_gemm_f32', L.ptr(dout), nc, 1, L.ptr(tape.feats), c_last, 1, None, L.ptr(new_like(cls.weight)), c_last,
                       nc, c_last, N, 0, st)
            if want(cls.bias):
                ws = _f32(L.query('gnx_bn_workspace', N, nc), dev)
                L.call('gnx_colsum', L.ptr(dout), nc, N, nc, L.ptr(new_like(cls.bias)), 0, L.ptr(ws), st)
        else:
            dfeats = dout

        # ---- the loss scale of this backward, on the device
        sf = tape.statsf
        s_last = sizes[-1]
        S2 = s_last * s_last
        # (the target exponent: 12 = 16x of headroom below the fp16 maximum; the training loops lower it after a step that
        #  overflowed and restore it after a run of clean steps - training._F16StepGuard)
        tgt = float(model.__dict__.get('f16_grad_target', 12.0))
        top = dfeats.abs().max() * sf[0][:c_last].abs().max() / S2
        e = torch.where(top > 0, torch.floor(tgt - torch.log2(top.clamp_min(1e-38))), torch.zeros_like(top)).clamp(-24.0, 60.0)
        s_val = torch.exp2(e)
        ls = torch.stack([s_val, 1.0 / s_val]).to(F32).contiguous()
        model.__dict__['f16_grad_scale'] = ls
        lp = L.ptr(ls)
        block_scales = [ls]                                  # one {s, 1/s} per dense block, last block first (see the transitions)
        ls_cur = ls
        model.__dict__['f16_grad_block_scales'] = block_scales

        # ---- tail: norm_final -> relu -> global average.  Every block buffer and block gradient is channel-blocked: (ld, bs)
        #      = (32, rows * 32) in the `_lb` entry points
        bufs = tape.bufs
        dbufs = [None] * len(bufs)
        dbufs[-1] = torch.empty_like(bufs[-1])
        bsl = bufs[-1].shape[1] * 32
        dgf, dbf = bn_out(model.features.norm_final)
        ws = _f32(L.query('gnx_tail_bwd_f16_workspace', N, c_last), dev)
        L.call('gnx_tail_bwd_f16_lb', L.ptr(dfeats), c_last, bufs[-1].data_ptr(), 32, bsl, dbufs[-1].data_ptr(), 32, bsl, N, c_last, S2,
               L.ptr(sf[0]), L.ptr(sf[1]), L.ptr(sf[2]), L.ptr(sf[3]), L.ptr(dgf), L.ptr(dbf), L.ptr(ws), lp, 0, fp, st)

        # ---- dense blocks, last to first
        for bi in range(len(model._blocks) - 1, -1, -1):
            c_in, layers, trans, c_total = model._blocks[bi]
            s = sizes[bi]
            M = N * s * s
            X, G = bufs[bi], dbufs[bi]
            bs = M * 32                                      # halves between two channel blocks of X, G and the taped A
            dB = torch.empty((M, mid), device=dev, dtype=H16)          # (scratch of this block's layers: row-major)
            ws3 = _f32(L.query('gnx_wgrad3x3_f16_workspace', M), dev)
            wsd3 = _f32(L.query('gnx_conv3x3_dgrad_bnrelu_bwd_f16_workspace', M), dev)
            wsc3 = _f32(L.query('gnx_conv3x3_bwd_f16_workspace', M), dev)
            for li in range(len(layers) - 1, -1, -1):
                layer = layers[li]
                a, s1, s2 = tape.layers[bi][li]
                cin = c_in + li * g
                dy = G.data_ptr() + 2 * (cin // 32) * bs     # the layer's 32 gradient columns: ONE contiguous [M][32] matrix
                w2 = layer.conv2.weight
                w1 = layer.conv1.weight
                # both fp16 operands of the layer's backward in one launch: W2b [tap][m][n], W1t [cin][128]
                w2b = torch.empty((9, mid, g), device=dev, dtype=H16)
                w1t = torch.empty((cin, mid), device=dev, dtype=H16)
                L.call('gnx_dense_bwd_f16_pack', L.ptr(w1.detach().contiguous()), L.ptr(w2.detach().contiguous()), w1t.data_ptr(),
                       w2b.data_ptr(), cin, st)                   # (mid = 128, g = 32: `eligible`)
                dg2, db2 = bn_out(layer.norm2)
                if want(w2) and model.f16_fused_conv2_backward:
                    # conv2's whole backward - data gradient + norm2 adjoint AND weight gradient - in ONE pass over dY and A
                    t0 = model._probe_begin()
                    L.call('gnx_conv3x3_bwd_f16_lb', dy, 32, w2b.data_ptr(), a.data_ptr(), 32, bs, dB.data_ptr(), L.ptr(new_like(w2)),
                           M, s, L.ptr(s2[0]), L.ptr(layer.norm2.weight), L.ptr(layer.norm2.bias), L.ptr(dg2), L.ptr(db2), L.ptr(wsc3),
                           lp, 0, fp, st)
                    model._probe_mark('conv3x3_bwd_f16', t0, 4 * M * 9 * mid * g, 2 * M * (g + 2 * mid))
                else:
                    if want(w2):
                        t0 = model._probe_begin()
                        L.call('gnx_wgrad3x3_f16_lb', dy, 32, a.data_ptr(), 32, bs, L.ptr(new_like(w2)), L.ptr(ws3), M, s, lp, 0, fp, st)
                        model._probe_mark('wgrad3x3_f16', t0, 2 * M * 9 * mid * g, 2 * M * (mid + g))
                    t0 = model._probe_begin()
                    L.call('gnx_conv3x3_dgrad_bnrelu_bwd_f16_lb', dy, 32, w2b.data_ptr(), a.data_ptr(), 32, bs, dB.data_ptr(), M, s,
                           L.ptr(s2[0]), L.ptr(layer.norm2.weight), L.ptr(layer.norm2.bias), L.ptr(dg2), L.ptr(db2), L.ptr(wsd3), lp, 0,
                           fp, st)
                    model._probe_mark('dgrad3x3_bn2_f16', t0, 2 * M * 9 * mid * g, 2 * M * (g + 2 * mid))
                # conv1: data gradient + norm1 -> relu1's adjoint into the block gradient, and - from the same staged tiles - the
                # weight gradient (ONE pass over dB, X and G)
                dg1, db1 = bn_out(layer.norm1)
                t0 = model._probe_begin()
                wg = want(w1)
                wsd1 = _f32(L.query('gnx_conv1x1_dgrad_wgrad_f16_workspace' if wg else 'gnx_conv1x1_dgrad_bnrelu_bwd_f16_workspace',
                                    M, cin), dev)
                L.call('gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16_lb', dB.data_ptr(), w1t.data_ptr(), X.data_ptr(), 32, bs, G.data_ptr(),
                       32, bs, M, cin, L.ptr(s1[0]), L.ptr(s1[1]), L.ptr(s1[2]), L.ptr(s1[3]), L.ptr(dg1), L.ptr(db1),
                       L.ptr(new_like(w1)) if wg else None, L.ptr(wsd1), lp, 0, fp, st)
                model._probe_mark('dgrad_wgrad1x1_bn1_f16' if wg else 'dgrad1x1_bn1_f16', t0, (4 if wg else 2) * M * cin * mid,
                                  2 * M * (mid + 3 * cin))
                tape.layers[bi][li] = None
                del a
            del dB, ws3, wsd3, wsc3
            if bi > 0:
                # transition bi-1 -> bi: its output gradient is the first c_out / 32 channel blocks of this block's gradient;
                # the two GEMMs that consume it read rows, so those blocks are copied out once as a row-major matrix
                p_c_in, p_layers, p_trans, p_total = model._blocks[bi - 1]
                ps = sizes[bi - 1]
                stt, pooled = tape.trans[bi - 1]
                c_out = p_trans.conv.out_channels
                wt = p_trans.conv.weight
                Gn = _rows_of_blocks(G, c_out // 32)                                                   # [M][c_out]
                if want(wt):
                    wsw = _f32(L.query('gnx_wgrad1x1_f16_workspace', M, c_out, p_total), dev)
                    L.call('gnx_wgrad1x1_f16', Gn.data_ptr(), c_out, pooled.data_ptr(), p_total, None, None, L.ptr(new_like(wt)),
                           L.ptr(wsw), M, c_out, p_total, lp, 0, fp, st)
                    del wsw
                tape.trans[bi - 1] = None
                del pooled
                wtt = wt.detach().reshape(c_out, p_total).t().to(H16).contiguous()                    # [p_total][c_out]
                dPool = torch.empty((p_total // 32, M, 32), device=dev, dtype=H16)                     # (channel-blocked)
                L.call('gnx_conv1x1_bnrelu_h16_cb', Gn.data_ptr(), c_out, wtt.data_ptr(), dPool.data_ptr(), M, M, p_total, c_out,
                       None, None, None, None, st)
                del Gn
                dbufs[bi - 1] = torch.empty_like(bufs[bi - 1])
                dgt, dbt = bn_out(p_trans.norm)
                wst = _f32(L.query('gnx_trans_bwd_f16_workspace', N, p_total, ps), dev)
                # Re-centre the scale for the block in front: gradients of an untrained network grow towards the input (x 4-16
                # per block measured), and a dense block adds up to 24 layers' contributions on top.  f = the power of two that
                # puts the largest pooled-gradient element back at 2^12; it rides on the transition norm's folded (scale, shift)
                # - the ReLU mask's sign test is unchanged by a positive factor, the BatchNorm sums do not use them - so the kernel
                # writes f x its block gradient while its own sums still carry the old scale.
                mn, mx = torch.aminmax(dPool)                   # (one pass; abs() would write a copy of the tensor first)
                amax = torch.maximum(mx, -mn).to(F32)
                f_e = torch.where(amax > 0, torch.floor(tgt - torch.log2(amax.clamp_min(1e-30))), torch.zeros_like(amax))
                f_e = torch.minimum(torch.maximum(f_e, -24.0 - torch.log2(ls_cur[0])), 60.0 - torch.log2(ls_cur[0])).clamp(-12.0, 12.0)
                f_val = torch.exp2(f_e)
                sc_f, sh_f = (stt[0] * f_val).contiguous(), (stt[1] * f_val).contiguous()      # (kept alive across the call)
                pbs = bufs[bi - 1].shape[1] * 32
                L.call('gnx_trans_bwd_f16_lb', dPool.data_ptr(), 32, bs, bufs[bi - 1].data_ptr(), 32, pbs, dbufs[bi - 1].data_ptr(),
                       32, pbs, N, p_total, ps, L.ptr(sc_f), L.ptr(sh_f), L.ptr(stt[2]), L.ptr(stt[3]), L.ptr(dgt), L.ptr(dbt),
                       L.ptr(wst), lp, 0, fp, st)
                s_new = ls_cur[0] * f_val
                ls_cur = torch.stack([s_new, 1.0 / s_new]).to(F32).contiguous()
                block_scales.append(ls_cur)
                lp = L.ptr(ls_cur)
                del dPool, wst
                dbufs[bi] = None
                bufs[bi] = None
            send_bucket()

        # ---- stem: the gradient of the pooled map = the first c0 / 32 channel blocks of block 1's gradient, as rows
        conv0 = model.features.conv0
        norm0 = model.features.norm0
        c0 = conv0.out_channels
        need0 = want(conv0.weight) or want(norm0.weight) or want(norm0.bias)
        Gs = _rows_of_blocks(dbufs[0], c0 // 32) if need0 else None                                    # [M1][c0]
        dbufs[0] = None
        if tape.pool_idx is None and need0:
            # one pass over the patches: conv0 rows recomputed, pool0's winners found, gradient routed and contracted
            t0 = model._probe_begin()
            ws = _f32(L.query('gnx_stem_bwd_f16_workspace', N, P), dev)
            s0 = tape.stats0
            dg0, db0 = bn_out(norm0)
            L.call('gnx_stem_bwd_f16', L.ptr(tape.x), L.ptr(conv0.weight.detach().contiguous()), L.ptr(s0[0]), L.ptr(s0[1]), L.ptr(norm0.weight),
                   L.ptr(norm0.bias), Gs.data_ptr(), c0,
                   L.ptr(new_like(conv0.weight)) if want(conv0.weight) else None, L.ptr(dg0), L.ptr(db0), L.ptr(ws), N, P, c0, lp, 0,
                   fp, st)
            model._probe_mark('stem_bwd_f16', t0, 4 * N * hs * hs * c0 * 147, 4 * N * 3 * P * P + 2 * N * (hs // 2) ** 2 * c0)
        elif need0:
            hp = (hs + 2 - 3) // 2 + 1
            M1, M0 = N * hp * hp, N * hs * hs
            dO = torch.empty((M1, c0), device=dev, dtype=F32)
            L.call('gnx_h16_cols_to_f32', Gs.data_ptr(), c0, L.ptr(dO), c0, M1, c0, lp, fp, st)
            s0 = tape.stats0
            dS = torch.empty((M0, c0), device=dev, dtype=F32)
            L.call('gnx_maxpool_bwd_argmax_bnrelu', tape.pool_idx.data_ptr(), L.ptr(dO), c0, L.ptr(tape.stem32), c0, L.ptr(s0[0]),
                   L.ptr(dS), c0, N, c0, hs, hs, st)
            dg0, db0 = bn_out(norm0)
            ws = _f32(L.query('gnx_bn_workspace', M1, c0), dev)
            L.call('gnx_bn_relu_bwd', L.ptr(dO), c0, L.ptr(tape.stem32), c0, None, c0, M1, c0, L.ptr(s0[0]), L.ptr(s0[1]),
                   L.ptr(s0[2]), L.ptr(s0[3]), L.ptr(dg0), L.ptr(db0), 2, 0, 0, 0, L.ptr(ws), st)
            if want(conv0.weight):
                ws = _f32(L.query('gnx_conv0_wgrad_workspace', N, P, P, c0, 7, 7, 2, 3), dev)
                L.call('gnx_conv0_wgrad', L.ptr(tape.x), L.ptr(dS), c0, L.ptr(new_like(conv0.weight)), L.ptr(ws), N, P, P, c0, 7, 7,
                       2, 3, 0, st)
        del Gs
        if reducer is not None:
            send_bucket()
            reducer.finish()
        ctx.tape = None
        out = [None, None]
        for p in model.parameters():
            out.append(grads.get(p))
        return tuple(out)
