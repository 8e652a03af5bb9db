"""GPU parity: every C-ABI kernel family against the CPU oracle / a plain fp32 torch reference on the same
seeded inputs.  Tolerances are fp32 round-off (different summation order), stated per test."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def close(a, b, rtol=1e-4, atol=1e-5, what=''):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return
    err = (a - b).abs().max().item()
    tol = atol + rtol * b.abs().max().item()
    assert err <= tol, "%s max abs err %.3e > tol %.3e" % (what, err, tol)


@pytest.fixture(scope='module')
def GF():
    from gridnext_amd import functional
    return functional


@pytest.fixture(scope='module')
def L():
    from gridnext_amd import _lib
    return _lib


# ----------------------------------------------------------------------------------------------- hex conv
@pytest.mark.parametrize("B,H,W,I,O,oddr", [(2, 7, 6, 5, 9, True), (1, 78, 64, 16, 32, True),
                                             (2, 9, 8, 32, 7, False), (1, 5, 4, 14, 32, True),
                                             (3, 1, 1, 3, 4, True), (1, 2, 3, 32, 32, False),
                                             # wider than one launch takes (64 x 64 forward / 32 x 32 weight gradient):
                                             # tiled over channel chunks - many classes, classify=False feature inputs
                                             (1, 9, 7, 40, 32, True), (2, 6, 5, 64, 64, True), (1, 7, 6, 150, 70, True),
                                             (1, 5, 6, 33, 130, False),
                                             # every contraction width of the MFMA forward / data-gradient form (8, 16, 32, 64),
                                             # ragged position tiles, both addressings
                                             (2, 7, 5, 8, 16, False), (1, 11, 3, 16, 8, True), (3, 5, 9, 8, 8, True)])
def test_hexconv_fwd_bwd(GF, B, H, W, I, O, oddr):
    from oracle import hexconv as ohex
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + I)
    x = torch.randn(B, I, H, W, generator=g)
    k0 = torch.randn(O, I, 3, 1, generator=g) * 0.3
    k1 = torch.randn(O, I, 2, 2, generator=g) * 0.3
    b = torch.randn(O, generator=g)
    dy = torch.randn(B, O, H, W, generator=g)
    xr, k0r, k1r, br = [t.clone().requires_grad_(True) for t in (x, k0, k1, b)]
    ref = ohex.hexconv_oddr(xr, k0r, k1r, br) if oddr else ohex.hexconv_gather(xr, k0r, k1r, br)
    ref.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    k0d, k1d, bd = [t.to(DEV).requires_grad_(True) for t in (k0, k1, b)]
    out = GF.hexconv(xd, k0d, k1d, bd, oddr)
    out.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    close(out.permute(0, 3, 1, 2), ref, what='y')
    close(xd.grad.permute(0, 3, 1, 2), xr.grad, what='dx')
    close(k0d.grad, k0r.grad, rtol=2e-4, what='dk0')
    close(k1d.grad, k1r.grad, rtol=2e-4, what='dk1')
    close(bd.grad, br.grad, rtol=2e-4, what='db')


def test_hexagdly_compatible_module(GF):
    import gridnext_amd.hexconv as hexagdly
    from oracle import hexconv as ohex
    torch.manual_seed(3)
    m = hexagdly.Conv2d(6, 10, kernel_size=1, stride=1, bias=True)
    assert set(dict(m.named_parameters())) == {'kernel0', 'kernel1', 'bias_tensor'}
    assert float(m.bias_tensor.detach()[0]) == pytest.approx(0.01)
    x = torch.randn(2, 6, 9, 7)
    ref = ohex.hexconv_subconv(x, m.kernel0.detach(), m.kernel1.detach(), m.bias_tensor.detach())
    close(m.to(DEV)(x.to(DEV)), ref)


# ----------------------------------------------------------------------------------------------- batch norm
def _away_from_relu_kink(x, weight, bias, eps, training, running_mean=None, running_var=None):
    """Nudge the few entries of x whose BatchNorm output lies within 1e-4 of zero: there the fp32 kernel and a float64 reference
    may take different sides of the ReLU, an O(dy) difference in dx at that element that says nothing about either."""
    x = x.clone()
    for _ in range(4):
        xd = x.double()
        if training:
            mu, var = xd.mean(0), xd.var(0, unbiased=False)
        else:
            mu, var = running_mean.double(), running_var.double()
        y = (xd - mu) / (var + eps).sqrt() * weight.double() + bias.double()
        near = y.abs() < 1e-4
        if not near.any():
            break
        x[near] += 0.05
    return x


@pytest.mark.parametrize("M,C,relu,training", [(300, 32, True, True), (4992, 100, True, True), (77, 50, False, True),
                                               (512, 32, True, False), (5, 3, True, True),
                                               # round 5: the forms with several workgroups per channel block (2048 < M <= 8192)
                                               (4992, 32, True, True), (8192, 512, True, True), (2049, 36, False, True),
                                               (8192, 64, True, False), (5000, 1024, True, True), (4992, 50, True, True)])
def test_bn_relu_fwd_bwd(GF, M, C, relu, training):
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 3 + 5            # mean >> 0: exercises the two-pass variance
    dy = torch.randn(M, C, generator=g)
    bn_ref = nn.BatchNorm1d(C)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn_ref.bias.copy_(torch.randn(C, generator=g))
        bn_ref.running_mean.copy_(torch.randn(C, generator=g))
        bn_ref.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    import copy
    if relu:
        x = _away_from_relu_kink(x, bn_ref.weight.detach(), bn_ref.bias.detach(), bn_ref.eps, training, bn_ref.running_mean,
                                 bn_ref.running_var)
    bn_hip = copy.deepcopy(bn_ref).to(DEV)
    # the reference in float64: torch's fp32 CPU BatchNorm backward depends on the thread count and is FAR from fp64 at large
    # sizes on one or two threads (8192 x 512: dx off by 0.23 of its range on 1-2 threads, 2e-7 on 8) - and importing
    # tests/test_host_logic.py, which a whole-suite run does, sets torch to one thread
    bn_ref = bn_ref.double()
    bn_ref.train(training)
    bn_hip.train(training)
    xr = x.double().requires_grad_(True)
    yr = bn_ref(xr)
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.double())
    xd = x.to(DEV).requires_grad_(True)
    yd = GF.batch_norm_relu(xd, bn_hip, relu)
    yd.backward(dy.to(DEV))
    sync = getattr(bn_hip, '_gnx_sync', None)
    assert sync is None or int(sync.abs().sum().item()) == 0, "the BatchNorm sync words were not left zero: %s" % sync.tolist()[:24]
    close(yd, yr, what='y')
    close(xd.grad, xr.grad, rtol=3e-4, what='dx')
    close(bn_hip.weight.grad, bn_ref.weight.grad, rtol=3e-4, atol=1e-4, what='dgamma')
    close(bn_hip.bias.grad, bn_ref.bias.grad, rtol=3e-4, atol=1e-4, what='dbeta')
    close(bn_hip.running_mean, bn_ref.running_mean, what='running_mean')
    close(bn_hip.running_var, bn_ref.running_var, what='running_var')
    assert int(bn_hip.num_batches_tracked) == int(bn_ref.num_batches_tracked)


# ----------------------------------------------------------------------------------------------- masked CE
@pytest.mark.parametrize("C", [1, 17, 64, 65, 300])
def test_masked_ce_any_class_count(GF, C):
    """The fused CE / softmax take any number of classes (r1 refused C > 64): against torch's own CE on the CPU."""
    g = torch.Generator().manual_seed(C)
    M = 777
    z = torch.randn(M, C, generator=g) * 3
    lab = torch.randint(0, C + 1, (M,), generator=g)                    # 0 = background
    zr = z.clone().requires_grad_(True)
    keep = lab > 0
    ref = nn.functional.cross_entropy(zr[keep], lab[keep] - 1) / 2
    ref.backward()
    zd = z.to(DEV).requires_grad_(True)
    loss, stats, preds = GF.masked_cross_entropy(zd, lab.to(DEV), 2)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    close(zd.grad, zr.grad, rtol=1e-4, atol=1e-9)
    assert int(stats[0]) == int(keep.sum())
    assert torch.equal(preds.cpu(), z.argmax(1))


def test_masked_ce_reference_maps(GF):
    g = load_golden('masked_ce_maynard')
    z = torch.from_numpy(g['logits'])                      # (7, 78, 64)
    lab = torch.from_numpy(g['labels'])
    rows = z.permute(1, 2, 0).reshape(-1, 7).contiguous().to(DEV)
    for accum in (1, 4):
        r = rows.clone().requires_grad_(True)
        loss, stats, preds = GF.masked_cross_entropy(r, lab.to(DEV), accum)
        loss.backward()
        assert abs(loss.item() - float(g['loss_accum%d' % accum])) < 1e-5          # north-star CE tolerance 1e-4
        grad = r.grad.reshape(78, 64, 7).permute(2, 0, 1)
        close(grad, g['grad_accum%d' % accum], rtol=1e-4, atol=1e-9)
        assert int(stats[0]) == int((lab > 0).sum())
    from oracle import masked_ce as oce
    _, p_ref, t_ref = oce.masked_ce(z.unsqueeze(0), lab.unsqueeze(0))
    fg = (lab.reshape(-1) > 0)
    assert torch.equal(preds.cpu()[fg], p_ref)
    assert int(stats[1]) == int((p_ref == t_ref).sum())


@pytest.mark.parametrize("M,C,base", [(1, 2, 1), (1000, 8, 1), (257, 5, 0), (4992, 8, 1)])
def test_masked_ce_random(GF, M, C, base):
    g = torch.Generator().manual_seed(M)
    z = torch.randn(M, C, generator=g) * 4
    lab = torch.randint(0, C + base, (M,), generator=g)
    if base == 1:
        lab[0] = 1
    zr = z.clone().requires_grad_(True)
    keep = lab >= base
    ref = F.cross_entropy(zr[keep], lab[keep] - base) / 3
    ref.backward()
    zd = z.to(DEV).requires_grad_(True)
    loss, stats, preds = GF.masked_cross_entropy(zd, lab.to(DEV), 3, label_base=base)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    close(zd.grad, zr.grad, rtol=1e-4, atol=1e-9)
    assert int(stats[0]) == int(keep.sum())
    assert torch.equal(preds.cpu(), z.argmax(1))


def test_masked_ce_no_foreground_is_nan_like_torch(GF):
    z = torch.randn(10, 4).to(DEV)
    loss, stats, _ = GF.masked_cross_entropy(z, torch.zeros(10, dtype=torch.long, device=DEV), 1)
    assert torch.isnan(loss).item() and int(stats[0]) == 0


# ----------------------------------------------------------------------------------------------- Linear / MLP
@pytest.mark.parametrize("M,K,N", [(128, 2000, 500), (4992, 100, 50), (77, 53, 9), (1, 7, 3), (300, 64, 130)])
def test_linear_rows(GF, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    F.linear(xr, wr, br).backward(dy)
    xd, wd, bd = [t.to(DEV).requires_grad_(True) for t in (x, w, b)]
    y = GF.linear(xd, wd, bd)
    y.backward(dy.to(DEV))
    close(y, F.linear(x, w, b), rtol=2e-5 * max(1, K ** 0.5), what='y')
    close(xd.grad, xr.grad, rtol=1e-4, what='dx')
    close(wd.grad, wr.grad, rtol=1e-4 * max(1, M ** 0.5 / 8), what='dw')
    close(bd.grad, br.grad, rtol=1e-4 * max(1, M ** 0.5 / 8), what='db')


def test_linear_count_grid_in_place(GF):
    # (B, genes, H*W) count grid consumed K-major == permute(0,2,1).reshape(-1, genes) rows
    g = torch.Generator().manual_seed(5)
    B, G, S, N = 2, 203, 78 * 64, 37
    x = torch.randint(0, 10, (B, G, S), generator=g).float()
    w, b = torch.randn(N, G, generator=g) * 0.05, torch.randn(N, generator=g)
    dy = torch.randn(B * S, N, generator=g)
    wr = w.clone().requires_grad_(True)
    ref = F.linear(x.permute(0, 2, 1).reshape(-1, G), wr, b)
    ref.backward(dy)
    wd = w.to(DEV).requires_grad_(True)
    y = GF.linear(x.to(DEV), wd, b.to(DEV), kmajor=True)
    y.backward(dy.to(DEV))
    close(y, ref, rtol=1e-4)
    close(wd.grad, wr.grad, rtol=2e-3)


@pytest.mark.parametrize("B,H,W", [(1, 8, 6), (1, 78, 64), (2, 9, 7)])
def test_hexconv_weight_gradients_batched_equal_the_single_calls(L, B, H, W):
    """gnx_hexconv_bwd_weight_batch (the corrector's five layers as one launch + one batched reduce, what a captured step runs)
    == gnx_hexconv_bwd_weight per layer, bit for bit: dkernel0, dkernel1 and the bias gradient; a layer wider than 32 channels
    is declined (UNSUPPORTED, nothing written)."""
    import ctypes
    from gridnext_amd import functional as GF
    layers = [(5, 32), (32, 32), (32, 32), (32, 32), (32, 5)]
    g = torch.Generator().manual_seed(H * 7 + W)
    single, batched, ops = [], [], []
    for (I, O) in layers:
        x = torch.randn(B, H, W, I, generator=g).to(DEV)
        dy = torch.randn(B, H, W, O, generator=g).to(DEV)
        outs = [(torch.full((O, I, 3, 1), 7.0, device=DEV), torch.full((O, I, 2, 2), 7.0, device=DEV), torch.full((O,), 7.0, device=DEV))
                for _ in range(2)]
        ws = torch.empty(L.query('gnx_hexconv_bwd_weight_workspace', B, H, W, I, O), device=DEV)
        L.call('gnx_hexconv_bwd_weight', L.ptr(x), L.ptr(dy), L.ptr(outs[0][0]), L.ptr(outs[0][1]), L.ptr(outs[0][2]), L.ptr(ws),
               B, H, W, I, O, 1, 0, L.stream())
        single.append(outs[0])
        batched.append(outs[1])
        ops.append((x, dy, I, O))
    arr = (GF._HexWgradItem * len(layers))()
    keep = []
    for a, (x, dy, I, O), (dk0, dk1, db) in zip(arr, ops, batched):
        ws = torch.empty(L.query('gnx_hexconv_bwd_weight_workspace', B, H, W, I, O), device=DEV)
        keep.append(ws)
        a.x, a.dy, a.dkernel0, a.dkernel1, a.dbias, a.workspace = L.ptr(x), L.ptr(dy), L.ptr(dk0), L.ptr(dk1), L.ptr(db), L.ptr(ws)
        a.B, a.H, a.W, a.I, a.O, a.mode, a.accumulate, a.pad = B, H, W, I, O, 1, 0, 0
    L.call('gnx_hexconv_bwd_weight_batch', ctypes.addressof(arr), len(layers), L.stream())
    torch.cuda.synchronize()
    for s_, b_ in zip(single, batched):
        for ts, tb in zip(s_, b_):
            assert torch.equal(ts, tb)
    arr[0].I = 40
    assert L.query('gnx_hexconv_bwd_weight_batch', ctypes.addressof(arr), len(layers), L.stream()) == L.ERR_UNSUPPORTED


@pytest.mark.parametrize("kmajor,S,K,N", [(True, 4992, 2000, 500), (True, 2308, 516, 260), (False, 2500, 1000, 384),
                                          (True, 4992, 2000, 100), (False, 2100, 1030, 70), (True, 2496, 1024, 128)])
def test_linear_whole_grid_split_k_form(GF, kmajor, S, K, N):
    """The first Linear of the count MLP over a whole grid (gemm_f32.hip: 256 x 128 tiles for N >= 256, 64 x 64 tiles for the
    tall, narrow composed 2000 -> 100 layer; K split over workgroups, slabs summed in a fixed order) against fp64: ragged
    tiles in every dimension, bias, twice the same bits."""
    from gridnext_amd import _lib as L
    assert L.query('gnx_gemm_f32_workspace', S, N, K) > 0
    g = torch.Generator().manual_seed(S + K + N)
    w, b = torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    if kmajor:
        x = torch.randint(0, 10, (1, K, S), generator=g).float()
        rows = x[0].t()
    else:
        x = torch.randn(S, K, generator=g)
        rows = x
    ref = rows.double() @ w.double().t() + b.double()
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y1 = GF.linear(xd, wd, bd, kmajor=kmajor)
    y2 = GF.linear(xd, wd, bd, kmajor=kmajor)
    assert torch.equal(y1, y2)
    err = (y1.double().cpu() - ref).abs().max().item()
    assert err <= 2e-6 * K ** 0.5 * ref.abs().max().item(), err


def test_count_mlp_pipeline_vs_reference_fixture(GF):
    from gridnext_amd.synthetic import count_mlp
    g = load_golden('mlp_count')
    f = count_mlp(200, 8)
    f.load_state_dict(sub(g, 'sd'))
    f.to(DEV)
    x = torch.from_numpy(g['x']).to(DEV)
    f.eval()
    close(GF.sequential_forward(f, x), g['eval_out'], rtol=1e-4, what='eval')
    f.train()
    xg = x.clone().requires_grad_(True)
    y = GF.sequential_forward(f, xg)
    close(y, g['train_out'], rtol=2e-4, what='train')
    loss, stats, _ = GF.masked_cross_entropy(y, torch.from_numpy(g['labels']).to(DEV), 1, label_base=0)
    loss.backward()
    assert abs(loss.item() - float(g['train_loss'])) < 1e-4
    close(xg.grad, g['traingrad/x'], rtol=1e-3, atol=1e-7)
    for k, p in f.named_parameters():
        close(p.grad, g['traingrad/' + k], rtol=2e-3, atol=1e-6, what=k)
    for k, ref in sub(g, 'post').items():
        close(f.state_dict()[k], ref, rtol=1e-4, what=k)


# ----------------------------------------------------------------------------------------------- DenseNet pieces
def _call_conv1x1(L, A, W, scale, shift, pool=0, S=0, ldc=None, col0=0):
    M_in, K = A.shape
    N = W.shape[0]
    M = M_in // 4 if pool else M_in
    ldc = ldc or N
    out = torch.zeros((M, ldc), device=DEV)
    L.call('gnx_conv1x1_bnrelu', L.ptr(A), A.stride(0), L.ptr(W), out.data_ptr() + 4 * col0, ldc, M, N, K,
           L.ptr(scale), L.ptr(shift), pool, S, L.stream())
    return out


@pytest.mark.parametrize("M,K,N,act", [(1024, 64, 128, True), (300, 22, 12, True), (129, 96, 128, False),
                                       (4096, 224, 128, True), (64, 1024, 512, True), (5, 3, 2, True),
                                       # whole 128 x 128 x 32 tiles: the LDS-DMA kernel (odd and even chunk counts)
                                       (256, 96, 256, False), (384, 32, 128, True), (128, 1024, 128, True),
                                       # ragged last column tile (32 | N): the data-gradient shapes N = channels-in
                                       (256, 128, 96, False), (128, 128, 160, True), (640, 128, 352, False)])
def test_conv1x1_bnrelu(L, M, K, N, act):
    g = torch.Generator().manual_seed(M + K)
    Afull = torch.randn(M, K + 8, generator=g)                   # leading dimension != K
    A = Afull[:, :K]
    W = torch.randn(N, K, generator=g) * 0.1
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    a = torch.relu(A * sc + sh) if act else A
    ref = a @ W.t()
    Ad = Afull.to(DEV)[:, :K]
    assert Ad.stride(0) == K + 8
    out = _call_conv1x1(L, Ad, W.to(DEV), sc.to(DEV) if act else None, sh.to(DEV) if act else None)
    close(out, ref, rtol=1e-4 * max(1, K ** 0.5 / 4))


@pytest.mark.parametrize("n,S,K,N", [(3, 8, 64, 32), (2, 7, 10, 5), (1, 32, 256, 128),
                                     # whole tiles: the wave-specialised kernel with pooling producers (tiles across
                                     # image boundaries, two column tiles, odd chunk count)
                                     (8, 8, 96, 256), (4, 16, 64, 128), (40, 4, 1024, 512)])
def test_transition_pool_first(L, n, S, K, N):
    g = torch.Generator().manual_seed(S + K)
    x = torch.randn(n, K, S, S, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    ref = F.avg_pool2d(F.conv2d(torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), W.view(N, K, 1, 1)), 2, 2)
    A = x.permute(0, 2, 3, 1).reshape(-1, K).contiguous().to(DEV)
    So = S // 2
    out = torch.zeros((n * So * So, N), device=DEV)
    Wd, scd, shd = W.to(DEV), sc.to(DEV), sh.to(DEV)          # keep device buffers alive across the async launch
    L.call('gnx_conv1x1_bnrelu', L.ptr(A), K, L.ptr(Wd), L.ptr(out), N, n * So * So, N, K,
           L.ptr(scd), L.ptr(shd), 1, S, L.stream())
    close(out.reshape(n, So, So, N).permute(0, 3, 1, 2), ref, rtol=2e-4)


@pytest.mark.parametrize("M,K,N", [(1024, 64, 128), (300, 22, 12), (4096, 224, 128), (129, 96, 130), (256, 64, 96)])
def test_conv1x1_bnrelu_act(L, M, K, N):
    """conv1 storing relu(bn2(.)): the eval forward's bottleneck, ready for the prologue-free conv3x3."""
    g = torch.Generator().manual_seed(M + K + 1)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    osc, osh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.5
    ref = torch.relu((torch.relu(A * sc + sh) @ W.t()) * osc + osh)
    Ad, Wd, scd, shd, oscd, oshd = (v.to(DEV) for v in (A, W, sc, sh, osc, osh))
    out = torch.full((M, N + 3), 7.0, device=DEV)
    L.call('gnx_conv1x1_bnrelu_act', L.ptr(Ad), K, L.ptr(Wd), L.ptr(out), N + 3, M, N, K, L.ptr(scd), L.ptr(shd),
           L.ptr(oscd), L.ptr(oshd), L.stream())
    close(out[:, :N], ref, rtol=2e-4)
    assert float(out[:, N:].min()) == 7.0
    assert L.lib().gnx_conv1x1_bnrelu_act(L.ptr(Ad), K, L.ptr(Wd), L.ptr(out), N + 3, M, N, K, L.ptr(scd), L.ptr(shd),
                                          None, None, L.stream()) != 0       # both output vectors are required


@pytest.mark.parametrize("M,K,lda,act", [(128, 64, 64, True), (1000, 96, 128, True), (333, 224, 256, False), (4096, 480, 512, True),
                                          (2050, 992, 1024, True), (77, 36, 40, True)])
def test_conv1x1_split_bf16_operands(L, M, K, lda, act):
    """Late round 5: conv1 (+ norm2 / relu2 on the store) with every fp32 operand split into two bf16 numbers and three matrix
    instructions per product (csrc/conv1x1_split.hip).  Against float64 on ragged M, K not a multiple of the 64-wide chunk, a wider
    operand row (block buffer): error of fp32 grade - gated at 3e-5 of the output range, where bf16 operands alone sit at 4e-3 and
    the fp32 instruction at 1e-6 - tail columns untouched, and the error reported beside the fp32 kernel's."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, lda, generator=g)
    W = torch.randn(128, K, generator=g) * (1.0 / K ** 0.5)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    osc, osh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.5
    y = torch.relu(A[:, :K].double() * sc.double() + sh.double()) @ W.double().t()
    ref = torch.relu(y * osc.double() + osh.double()) if act else y
    Ad, Wd, scd, shd, oscd, oshd = (v.to(DEV) for v in (A, W, sc, sh, osc, osh))
    Wp = torch.empty(L.query('gnx_conv1x1_split_pack_halves', K), device=DEV, dtype=torch.bfloat16)
    L.call('gnx_conv1x1_split_pack', L.ptr(Wd), Wp.data_ptr(), K, L.stream())
    out = torch.full((M, 131), 7.0, device=DEV)
    L.call('gnx_conv1x1_bnrelu_act_split', L.ptr(Ad), lda, Wp.data_ptr(), L.ptr(out), 131, M, K, L.ptr(scd), L.ptr(shd),
           L.ptr(oscd) if act else None, L.ptr(oshd) if act else None, L.stream())
    rng = ref.abs().max().item()
    err = (out[:, :128].double().cpu() - ref).abs().max().item()
    assert err <= 3e-5 * rng, (err, rng)
    assert float(out[:, 128:].min()) == 7.0 and float(out[:, 128:].max()) == 7.0
    if act:
        o32 = torch.empty(M, 128, device=DEV)
        L.call('gnx_conv1x1_bnrelu_act', L.ptr(Ad), lda, L.ptr(Wd), L.ptr(o32), 128, M, 128, K, L.ptr(scd), L.ptr(shd), L.ptr(oscd),
               L.ptr(oshd), L.stream())
        e32 = (o32.double().cpu() - ref).abs().max().item()
        print("\n[conv1x1 split bf16, M=%d K=%d] max error / range: split %.2e, fp32 instruction %.2e" % (M, K, err / rng, e32 / rng))
    # unsupported operand layouts are declined, not computed differently
    assert L.lib().gnx_conv1x1_bnrelu_act_split(L.ptr(Ad), lda + 1, Wp.data_ptr(), L.ptr(out), 131, M, K, L.ptr(scd), L.ptr(shd), None,
                                                None, L.stream()) == -3


@pytest.mark.parametrize("n,S,ldc", [(3, 4, 40), (5, 8, 32), (3, 16, 96), (2, 32, 64), (1, 64, 32), (37, 4, 32), (9, 8, 48)])
def test_conv3x3_split_bf16_operands(L, n, S, ldc):
    """Late round 5: conv2 (3x3, 128 -> 32, zero padding) as nine shifted products of split bf16 operands
    (csrc/conv3x3_split.hip).  Against float64 conv2d on whole maps of every size, tiles that start and end inside maps (n S S not
    a multiple of 256), an output that is a column range of a wider buffer: error of fp32 grade (gate 3e-5 of the range), the
    buffer's other columns untouched."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(100 * n + S)
    M = n * S * S
    A = torch.relu(torch.randn(M, 128, generator=g))                      # (an activated bottleneck)
    W = torch.randn(32, 128, 3, 3, generator=g) * (1.0 / 1152 ** 0.5)
    ref = F.conv2d(A.double().reshape(n, S, S, 128).permute(0, 3, 1, 2), W.double(), padding=1).permute(0, 2, 3, 1).reshape(M, 32)
    Ad, Wd = A.to(DEV), W.to(DEV)
    Wp = torch.empty(L.query('gnx_conv3x3_split_pack_halves'), device=DEV, dtype=torch.bfloat16)
    L.call('gnx_conv3x3_split_pack', L.ptr(Wd), Wp.data_ptr(), L.stream())
    out = torch.full((M, ldc + 5), 7.0, device=DEV)
    L.call('gnx_conv3x3_split', L.ptr(Ad), 128, Wp.data_ptr(), out.data_ptr() + 4 * 4, ldc + 5, M, S, L.stream())
    got = out[:, 4:36].double().cpu()
    rng = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= 3e-5 * rng, (err, rng)
    assert float(out[:, :4].min()) == 7.0 and float(out[:, 36:].min()) == 7.0 and float(out[:, 36:].max()) == 7.0
    print("\n[conv3x3 split bf16, n=%d S=%d] max error / range %.2e" % (n, S, err / rng))
    assert L.lib().gnx_conv3x3_split(L.ptr(Ad), 128, Wp.data_ptr(), L.ptr(out), ldc + 5, M - 1, S, L.stream()) == -3


@pytest.mark.parametrize("S", [32, 8])
def test_conv3x3_split_wide_and_narrow_tiles_agree_bit_for_bit(L, S):
    """gnx_conv3x3_split gives a wave 64 pixels (weight fragments shared between its two blocks) once the matrix holds 1 024 tiles of
    512 pixels, 32 pixels below that.  A pixel's sums are formed in the same order either way: the large matrix in one call equals
    its two halves in two calls bit for bit (so a model's chunk size never changes a result), and its first map equals float64."""
    import torch.nn.functional as F
    n = 512 * 1024 // (S * S)
    M = n * S * S
    g = torch.Generator(device=DEV).manual_seed(S)
    A = torch.relu(torch.randn(M, 128, device=DEV, generator=g))
    W = torch.randn(32, 128, 3, 3, device=DEV, generator=g) * (1.0 / 1152 ** 0.5)
    Wp = torch.empty(L.query('gnx_conv3x3_split_pack_halves'), device=DEV, dtype=torch.bfloat16)
    L.call('gnx_conv3x3_split_pack', L.ptr(W), Wp.data_ptr(), L.stream())
    whole = torch.empty(M, 32, device=DEV)
    L.call('gnx_conv3x3_split', L.ptr(A), 128, Wp.data_ptr(), L.ptr(whole), 32, M, S, L.stream())
    halves = torch.empty(M, 32, device=DEV)
    for k in range(2):
        L.call('gnx_conv3x3_split', A.data_ptr() + k * (M // 2) * 128 * 4, 128, Wp.data_ptr(), halves.data_ptr() + k * (M // 2) * 32 * 4, 32,
               M // 2, S, L.stream())
    assert torch.equal(whole, halves)
    ref = F.conv2d(A[:S * S].double().cpu().reshape(1, S, S, 128).permute(0, 3, 1, 2), W.double().cpu(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(S * S, 32)
    assert (whole[:S * S].double().cpu() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K,ldx,act,accumulate", [(1000, 128, 96, 128, True, 0), (4096, 128, 480, 512, True, 1),
                                                         (777, 128, 992, 1024, True, 0), (300, 128, 64, 64, False, 0),
                                                         (70000, 128, 160, 256, True, 0), (2000, 256, 512, 512, False, 0),
                                                         (1111, 512, 1024, 1024, False, 1), (900, 64, 128, 128, False, 0)])
def test_wgrad1x1_split_bf16_operands(L, M, N, K, ldx, act, accumulate):
    """Late round 5: conv1's weight gradient dW = dY^T relu(scale X + shift) on split bf16 operands (csrc/wgrad_split.hip: fp32
    operands in HBM, transposing LDS reads, three matrix instructions per product, slabs summed in a fixed order) against float64:
    ragged M (rows beyond it contribute nothing), K not a multiple of the 128-wide block, a wider X row, accumulation into an
    existing gradient, no activation, the transitions' shapes (N = 64 .. 512 output channels: several 128-wide blocks of n).  Error of
    fp32 grade: gate 3e-5 of the result's range."""
    g = torch.Generator().manual_seed(M + K)
    dY = torch.randn(M, N, generator=g)
    X = torch.randn(M, ldx, generator=g)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    ax = torch.relu(X[:, :K].double() * sc.double() + sh.double()) if act else X[:, :K].double()
    ref = dY.double().t() @ ax
    prior = torch.randn(N, K, generator=g)
    if accumulate:
        ref = ref + prior.double()
    dYd, Xd, scd, shd = (v.to(DEV) for v in (dY, X, sc, sh))
    dW = prior.to(DEV).clone()
    ws = torch.empty(L.query('gnx_wgrad1x1_split_workspace', M, N, K), device=DEV)
    L.call('gnx_wgrad1x1_split', L.ptr(dYd), N, L.ptr(Xd), ldx, L.ptr(scd) if act else None, L.ptr(shd) if act else None, L.ptr(dW),
           L.ptr(ws), M, N, K, accumulate, L.stream())
    rng = ref.abs().max().item()
    err = (dW.double().cpu() - ref).abs().max().item()
    print("\n[wgrad1x1 split bf16, M=%d K=%d] max error / range %.2e" % (M, K, err / rng))
    assert err <= 3e-5 * rng, (err, rng)
    dW2 = prior.to(DEV).clone()
    L.call('gnx_wgrad1x1_split', L.ptr(dYd), N, L.ptr(Xd), ldx, L.ptr(scd) if act else None, L.ptr(shd) if act else None, L.ptr(dW2),
           L.ptr(ws), M, N, K, accumulate, L.stream())
    assert torch.equal(dW, dW2)                                           # fixed summation order: the same bits every time


@pytest.mark.parametrize("n,S,lddy,accumulate", [(5, 4, 32, 0), (3, 8, 64, 1), (3, 16, 96, 0), (2, 32, 32, 0), (1, 64, 160, 0), (40, 8, 32, 0)])
def test_wgrad3x3_split_bf16_operands(L, n, S, lddy, accumulate):
    """Late round 5: conv2's weight gradient on split bf16 operands (csrc/wgrad_split.hip: the dY strip of a 64-pixel tile staged
    once and read at nine row offsets by transposing reads, out-of-map partners redirected to a row of zeros) against float64
    autograd: every map size, tiles that straddle maps, dY as a column range of a wider buffer, accumulation.  Gate 3e-5 of the
    range; the same bits on a second call."""
    g = torch.Generator().manual_seed(31 * n + S)
    M = n * S * S
    A = torch.relu(torch.randn(M, 128, generator=g))
    dYw = torch.randn(M, lddy, generator=g)
    c0 = lddy - 32
    x = A.double().reshape(n, S, S, 128).permute(0, 3, 1, 2).contiguous()
    gy = dYw[:, c0:].double().reshape(n, S, S, 32).permute(0, 3, 1, 2).contiguous()
    ref = torch.nn.grad.conv2d_weight(x, (32, 128, 3, 3), gy, padding=1)
    prior = torch.randn(32, 128, 3, 3, generator=g)
    if accumulate:
        ref = ref + prior.double()
    Ad, dYd = A.to(DEV), dYw.to(DEV)
    dW = prior.to(DEV).clone()
    ws = torch.empty(L.query('gnx_wgrad3x3_split_workspace', M), device=DEV)
    args = (dYd.data_ptr() + 4 * c0, lddy, L.ptr(Ad), 128, L.ptr(dW), L.ptr(ws), M, S, accumulate, L.stream())
    L.call('gnx_wgrad3x3_split', *args)
    rng = ref.abs().max().item()
    err = (dW.double().cpu() - ref).abs().max().item()
    print("\n[wgrad3x3 split bf16, n=%d S=%d] max error / range %.2e" % (n, S, err / rng))
    assert err <= 3e-5 * rng, (err, rng)
    dW2 = prior.to(DEV).clone()
    L.call('gnx_wgrad3x3_split', dYd.data_ptr() + 4 * c0, lddy, L.ptr(Ad), 128, L.ptr(dW2), L.ptr(ws), M, S, accumulate, L.stream())
    assert torch.equal(dW, dW2)


def test_frozen_count_mlp_composed_into_affine_stages(GF):
    """A frozen count MLP in eval mode (train_gridwise's tutorial recipe, training.py:126) evaluates as three affine stages -
    Linear -> Linear -> BatchNorm1d composed into one map each (there is no activation between the paired Linears,
    Tutorial_visium_count.ipynb cell 12): equal to the layer-by-layer evaluation and to fp64 torch to fp32 round-off, on rows and
    on a K-major count grid; not used once a parameter trains, in train mode, or with `fold_frozen = False`; recomposed when a
    parameter changes."""
    from gridnext_amd.synthetic import count_mlp
    torch.manual_seed(3)
    G, C = 2000, 8
    f = count_mlp(G, C).to(DEV)
    for bn in (m for m in f if isinstance(m, nn.BatchNorm1d)):
        bn.running_mean.normal_(0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_(0, 0.3)
    f.eval()
    for p in f.parameters():
        p.requires_grad = False
    x = torch.randint(0, 10, (4992, G), device=DEV).float()
    ref = f.double()(x.double()).float()
    f.float()
    y = GF.sequential_forward(f, x)
    assert len(f.__dict__['_gnx_affine_plan'][1]) == 3
    close(y, ref, rtol=2e-5, what='composed vs fp64')
    f.fold_frozen = False
    y_layers = GF.sequential_forward(f, x)
    f.fold_frozen = True
    close(y, y_layers, rtol=2e-5, what='composed vs layer by layer')
    grid = x.t().reshape(1, G, 78, 64).contiguous()                     # K-major count grid, read in place
    close(GF.sequential_forward(f, grid.reshape(1, G, -1), kmajor=True), y, rtol=1e-6, what='K-major')
    with torch.no_grad():
        f[0].weight.mul_(1.5)                                           # a changed parameter: recomposed
    close(GF.sequential_forward(f, x), f.double()(x.double()).float(), rtol=2e-5, what='after a change')
    f.float()
    for p in f.parameters():
        p.requires_grad = True
    out = GF.sequential_forward(f, x[:256])                             # trainable: the layer-by-layer path with a tape
    assert out.requires_grad
    with torch.no_grad():                                               # ... also in its no-grad (validation) passes
        assert torch.equal(GF.sequential_forward(f, x[:256]), out.detach())


@pytest.mark.parametrize("M,N,ld", [(128, 64, 64), (992, 96, 256), (128 * 600, 224, 256), (2048, 992, 1024), (40000, 160, 160)])
def test_conv1x1_dgrad_and_wgrad_in_one_pass(L, M, N, ld):
    """gnx_conv1x1_dgrad_wgrad_bnrelu_bwd (round 4): conv1's data gradient + norm1/relu1 adjoint + BatchNorm sums + conv1's
    WEIGHT gradient from one pass over dB, X and G == gnx_conv1x1_bnrelu + gnx_bn_relu_bwd (data side) and
    gnx_wgrad_bnrelu(taps = 1) (weight side) on the same inputs, and fp64 torch arithmetic; whole 32-row tiles (others:
    UNSUPPORTED), several tiles per workgroup, partial channel blocks."""
    K = 128
    g = torch.Generator().manual_seed(M + N)
    dY = torch.randn(M, K, generator=g).to(DEV)
    W1 = (torch.randn(K, N, generator=g) * 0.1).to(DEV)                  # conv1.weight [mid][cin]
    Wt = W1.t().contiguous()
    X = torch.randn(M, ld, generator=g).to(DEV)
    dX0 = torch.randn(M, ld, generator=g).to(DEV)
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.3
    gamma[1::6] *= -1.0
    mean, var = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    inv = 1.0 / torch.sqrt(var + 1e-5)
    sc, sh = (gamma * inv).to(DEV), (beta - mean * gamma * inv).to(DEV)
    meand, invd = mean.to(DEV), inv.to(DEV)
    # the separate kernels
    tC = torch.empty(M, ld, device=DEV)
    L.call('gnx_conv1x1_bnrelu', L.ptr(dY), K, L.ptr(Wt), L.ptr(tC), ld, M, N, K, None, None, 0, 0, L.stream())
    dX_ref = dX0.clone()
    dg_ref, db_ref = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    ws = torch.empty(L.query('gnx_bn_workspace', M, N), device=DEV)
    L.call('gnx_bn_relu_bwd', L.ptr(tC), ld, L.ptr(X), ld, L.ptr(dX_ref), ld, M, N, L.ptr(sc), L.ptr(sh), L.ptr(meand),
           L.ptr(invd), L.ptr(dg_ref), L.ptr(db_ref), 1, 0, 0, 1, L.ptr(ws), L.stream())
    dW_ref = torch.empty(K, N, device=DEV)
    wsw = torch.empty(L.query('gnx_wgrad_workspace', M, K, N, 1), device=DEV)
    L.call('gnx_wgrad_bnrelu', L.ptr(dY), K, L.ptr(X), ld, L.ptr(sc), L.ptr(sh), L.ptr(dW_ref), L.ptr(wsw), M, K, N, 0, 1, 0, 0,
           L.stream())
    # one pass
    dX = dX0.clone()
    dg, db, dW = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.full((K, N), 5.0, device=DEV)
    ws2 = torch.empty(L.query('gnx_conv1x1_dgrad_wgrad_workspace', M, N), device=DEV)
    assert L.query('gnx_conv1x1_dgrad_wgrad_bnrelu_bwd', L.ptr(dY), K, L.ptr(Wt), L.ptr(X), ld, L.ptr(dX), ld, M - 8, N, L.ptr(sc),
                   L.ptr(sh), L.ptr(meand), L.ptr(invd), L.ptr(dg), L.ptr(db), L.ptr(dW), L.ptr(ws2), 0, L.stream()) == L.ERR_UNSUPPORTED
    L.call('gnx_conv1x1_dgrad_wgrad_bnrelu_bwd', L.ptr(dY), K, L.ptr(Wt), L.ptr(X), ld, L.ptr(dX), ld, M, N, L.ptr(sc), L.ptr(sh),
           L.ptr(meand), L.ptr(invd), L.ptr(dg), L.ptr(db), L.ptr(dW), L.ptr(ws2), 0, L.stream())
    close(dX[:, :N], dX_ref[:, :N], rtol=1e-5)
    assert torch.equal(dX[:, N:], dX0[:, N:])                           # columns past cin untouched
    close(dg, dg_ref, rtol=2e-5, atol=1e-3)
    close(db, db_ref, rtol=2e-5, atol=1e-3)
    close(dW, dW_ref, rtol=2e-5, atol=1e-4)
    act = torch.relu(X[:, :N].double() * sc.double() + sh.double())
    close(dW, dY.double().t() @ act, rtol=1e-5, atol=1e-4, what='dW vs fp64')
    # accumulate
    L.call('gnx_conv1x1_dgrad_wgrad_bnrelu_bwd', L.ptr(dY), K, L.ptr(Wt), L.ptr(X), ld, L.ptr(dX0.clone()), ld, M, N, L.ptr(sc),
           L.ptr(sh), L.ptr(meand), L.ptr(invd), L.ptr(dg), L.ptr(db), L.ptr(dW), L.ptr(ws2), 1, L.stream())
    close(dW, 2 * dW_ref, rtol=2e-5, atol=2e-4)
    close(db, 2 * db_ref, rtol=2e-5, atol=2e-3)


@pytest.mark.parametrize("M,N,ld", [(128, 64, 64), (1024, 96, 256), (128 * 600, 224, 256), (2048, 992, 1024)])
def test_conv1x1_dgrad_fused_with_bn_relu_backward(L, M, N, ld):
    """conv1's data gradient with norm1 -> relu1's backward in its store == the two separate kernels (dX accumulated into the
    block gradient, dgamma, dbeta), eval statistics; ragged shapes must say UNSUPPORTED."""
    K = 128
    g = torch.Generator().manual_seed(M + N)
    dY = torch.randn(M, K, generator=g).to(DEV)
    W1 = (torch.randn(K, N, generator=g) * 0.1).to(DEV)                  # conv1.weight [mid][cin]
    Wt = W1.t().contiguous()
    X = torch.randn(M, ld, generator=g).to(DEV)
    dX0 = torch.randn(M, ld, generator=g).to(DEV)
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.3
    gamma[1::6] *= -1.0
    mean, var = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    inv = 1.0 / torch.sqrt(var + 1e-5)
    sc, sh = (gamma * inv).to(DEV), (beta - mean * gamma * inv).to(DEV)
    meand, invd = mean.to(DEV), inv.to(DEV)
    # the two-kernel form
    tC = torch.empty(M, ld, device=DEV)
    L.call('gnx_conv1x1_bnrelu', L.ptr(dY), K, L.ptr(Wt), L.ptr(tC), ld, M, N, K, None, None, 0, 0, L.stream())
    dX_ref = dX0.clone()
    dg_ref, db_ref = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    ws = torch.empty(L.query('gnx_bn_workspace', M, N), device=DEV)
    L.call('gnx_bn_relu_bwd', L.ptr(tC), ld, L.ptr(X), ld, L.ptr(dX_ref), ld, M, N, L.ptr(sc), L.ptr(sh), L.ptr(meand),
           L.ptr(invd), L.ptr(dg_ref), L.ptr(db_ref), 1, 0, 0, 1, L.ptr(ws), L.stream())
    # fused
    dX = dX0.clone()
    dg, db = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    ws2 = torch.empty(L.query('gnx_conv1x1_dgrad_bn_workspace', M, N), device=DEV)
    L.call('gnx_conv1x1_dgrad_bnrelu_bwd', L.ptr(dY), K, L.ptr(Wt), L.ptr(X), ld, L.ptr(dX), ld, M, N, K, L.ptr(sc),
           L.ptr(sh), L.ptr(meand), L.ptr(invd), L.ptr(dg), L.ptr(db), 0, L.ptr(ws2), L.stream())
    close(dX[:, :N], dX_ref[:, :N], rtol=1e-5)
    assert torch.equal(dX[:, N:], dX0[:, N:])                           # columns past cin untouched
    close(dg, dg_ref, rtol=2e-5, atol=1e-3)
    close(db, db_ref, rtol=2e-5, atol=1e-3)
    # against torch autograd on a slice (double)
    m = min(M, 512)
    xs = X[:m, :N].double().cpu().requires_grad_(True)
    z = torch.relu(xs * sc.double().cpu() + sh.double().cpu()) @ W1.double().cpu().t()
    z.backward(dY[:m].double().cpu())
    close(dX[:m, :N] - dX0[:m, :N], xs.grad, rtol=1e-4)
    assert L.query('gnx_conv1x1_dgrad_bnrelu_bwd', L.ptr(dY), K, L.ptr(Wt), L.ptr(X), ld, L.ptr(dX), ld, M - 64, N, K,
                   L.ptr(sc), L.ptr(sh), L.ptr(meand), L.ptr(invd), L.ptr(dg), L.ptr(db), 0, L.ptr(ws2),
                   L.stream()) == L.ERR_UNSUPPORTED


@pytest.mark.parametrize("M,K,N,lda", [(128, 32, 128, 32), (1024, 64, 128, 256), (4096, 224, 128, 256), (128 * 700, 96, 128, 128),
                                       (2048, 992, 128, 1024), (1280, 160, 256, 160)])
def test_conv1x1_clamped_act(L, M, K, N, lda):
    """norm1 folded into conv1's operands, the ReLU done as a clamp inside the LDS: == relu(bn2(conv1(relu(bn1(x))))) up to
    rounding, with negative, zero and denormal-small scales among the channels; ragged shapes must say UNSUPPORTED."""
    g = torch.Generator().manual_seed(M + K + 2)
    A = torch.randn(M, lda, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    sc[1::5] *= -1.0                                           # gamma < 0: the clamp is a min
    sc[2::7] = 0.0                                             # gamma = 0: the channel is the constant relu(shift)
    sc[3] = 1e-42                                              # -shift/scale overflows: treated like gamma = 0
    osc, osh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.5
    ref = torch.relu((torch.relu(A[:, :K].double() * sc.double() + sh.double()) @ W.double().t()) * osc.double() + osh.double())
    Ad, Wd, scd, shd, oscd, oshd = (v.to(DEV) for v in (A, W, sc, sh, osc, osh))
    Wf, bounds, oshf = torch.empty(N, K, device=DEV), torch.empty(2, K, device=DEV), torch.empty(N, device=DEV)
    L.call('gnx_conv1x1_fold_clamp', L.ptr(Wd), L.ptr(scd), L.ptr(shd), L.ptr(oscd), L.ptr(oshd), L.ptr(Wf), L.ptr(bounds),
           L.ptr(oshf), N, K, L.stream())
    out = torch.full((M, N + 4), 7.0, device=DEV)
    L.call('gnx_conv1x1_clamped_act', L.ptr(Ad), lda, L.ptr(Wf), L.ptr(bounds), L.ptr(out), N + 4, M, N, K, L.ptr(oscd),
           L.ptr(oshf), L.stream())
    close(out[:, :N].double(), ref, rtol=2e-4)
    assert float(out[:, N:].min()) == 7.0
    # same answer as the unfolded kernel, to rounding
    out2 = torch.empty((M, N), device=DEV)
    L.call('gnx_conv1x1_bnrelu_act', L.ptr(Ad), lda, L.ptr(Wd), L.ptr(out2), N, M, N, K, L.ptr(scd), L.ptr(shd),
           L.ptr(oscd), L.ptr(oshd), L.stream())
    close(out[:, :N], out2, rtol=2e-4)
    for m, n, k in ((M - 64, N, K), (M, N - 32, K), (M, N, K - 16)):
        assert L.query('gnx_conv1x1_clamped_act', L.ptr(Ad), lda, L.ptr(Wf), L.ptr(bounds), L.ptr(out), N + 4, m, n, k,
                       L.ptr(oscd), L.ptr(oshf), L.stream()) == L.ERR_UNSUPPORTED


@pytest.mark.parametrize("n,S,K,N,act", [(2, 8, 128, 32, True), (3, 4, 12, 6, True), (1, 32, 128, 32, True),
                                         (5, 7, 16, 4, False), (2, 14, 128, 32, True), (1, 56, 8, 4, True),
                                         (33, 4, 128, 32, True), (1, 1, 8, 4, True), (2, 2, 128, 32, True),
                                         # prologue-free inputs, 128 | M, N = 32, 32 | K: the LDS-DMA persistent kernel
                                         (2, 8, 128, 32, False), (8, 16, 128, 32, False), (3, 32, 64, 32, False),
                                         (1, 64, 32, 32, False), (600, 4, 96, 32, False), (40, 32, 128, 32, False),
                                         (128, 7, 160, 32, False),
                                         # data-gradient shape (K = 32 in, N = 64 / 128 out): column tiles as chunks
                                         (2, 8, 32, 128, False), (8, 16, 32, 64, False), (40, 32, 32, 128, False),
                                         (256, 32, 32, 128, False), (64, 7, 32, 128, False)])
def test_conv3x3_bnrelu(L, n, S, K, N, act):
    g = torch.Generator().manual_seed(S * 100 + K)
    x = torch.randn(n, K, S, S, generator=g)
    W = torch.randn(N, K, 3, 3, generator=g) * 0.1
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if act else x
    ref = F.conv2d(a, W, padding=1)
    A = x.permute(0, 2, 3, 1).reshape(-1, K).contiguous().to(DEV)
    Wd = W.to(DEV)
    Wr = torch.empty((9, N, K), device=DEV)
    L.call('gnx_repack_conv3x3', L.ptr(Wd), L.ptr(Wr), N, K, L.stream())
    ldc = N + 5
    out = torch.full((n * S * S, ldc), 7.0, device=DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    L.call('gnx_conv3x3_bnrelu', L.ptr(A), K, L.ptr(Wr), out.data_ptr() + 4 * 3, ldc, n * S * S, N, K, S,
           L.ptr(scd) if act else None, L.ptr(shd) if act else None, L.stream())
    close(out[:, 3:3 + N].reshape(n, S, S, N).permute(0, 3, 1, 2), ref, rtol=2e-4)
    assert float(out[:, :3].min()) == 7.0 and float(out[:, 3 + N:].min()) == 7.0     # neighbours untouched


@pytest.mark.parametrize("M,K,N", [(1024, 64, 128), (300, 24, 16), (4096, 224, 128), (77, 96, 200)])
def test_conv1x1_f16_act16(L, M, K, N):
    """Config 5: conv1 with fp16 MFMA operands storing the bottleneck activated and rounded to fp16."""
    g = torch.Generator().manual_seed(M + K + 3)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    osc, osh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.5
    act = torch.relu(A * sc + sh).half().double()                  # operands as the kernel rounds them
    ref = torch.relu((act @ W.half().double().t()) * osc.double() + osh.double())
    Ad, Wd, scd, shd, oscd, oshd = (v.to(DEV) for v in (A, W, sc, sh, osc, osh))
    ldc = (N + 15) // 8 * 8
    out = torch.full((M, ldc), 7.0, device=DEV, dtype=torch.float16)
    L.call('gnx_conv1x1_bnrelu_f16_act16', L.ptr(Ad), K, L.ptr(Wd), L.ptr(out, torch.float16), ldc, M, N, K, L.ptr(scd), L.ptr(shd),
           L.ptr(oscd), L.ptr(oshd), L.stream())
    close(out[:, :N].double(), ref, rtol=1.5e-3)                   # fp16 rounding of the stored value: 2^-11
    assert float(out[:, N:].float().min()) == 7.0


@pytest.mark.parametrize("n,S", [(8, 4), (2, 8), (1, 16), (40, 32), (4, 64), (256, 32), (4992, 4)])
def test_conv3x3_f16_dma(L, n, S):
    """Config 5: conv2 on the fp16 bottleneck, both operands global -> LDS by DMA, v_mfma_f32_32x32x16_f16, fp32
    accumulation: equal to the convolution of the same fp16 values up to fp32 summation order."""
    K, N = 128, 32
    g = torch.Generator().manual_seed(n * 100 + S)
    x16 = torch.relu(torch.randn(n, K, S, S, generator=g)).half()
    W = torch.randn(N, K, 3, 3, generator=g) * 0.05
    Wd = W.to(DEV)
    Wr = torch.empty(9, N, K, device=DEV)
    L.call('gnx_repack_conv3x3', L.ptr(Wd), L.ptr(Wr), N, K, L.stream())
    Wr16 = Wr.half()
    if n * S * S <= 65536:
        ref = F.conv2d(x16.double(), W.half().double(), padding=1)
    else:                                                          # the large cases: fp32 reference on the GPU
        ref = F.conv2d(x16.to(DEV).float(), W.half().float().to(DEV), padding=1).cpu()
    A16 = x16.permute(0, 2, 3, 1).reshape(-1, K).contiguous().to(DEV)
    ldc = N + 8
    out = torch.full((n * S * S, ldc), 7.0, device=DEV)
    L.call('gnx_conv3x3_f16_dma', L.ptr(A16, torch.float16), K, L.ptr(Wr16, torch.float16), out.data_ptr() + 4 * 4, ldc, n * S * S, N, K, S, L.stream())
    close(out[:, 4:4 + N].reshape(n, S, S, N).permute(0, 3, 1, 2), ref, rtol=2e-4)
    assert float(out[:, :4].min()) == 7.0 and float(out[:, 4 + N:].min()) == 7.0
    assert L.query('gnx_conv3x3_f16_dma', L.ptr(A16, torch.float16), K, L.ptr(Wr16, torch.float16), L.ptr(out), ldc, n * S * S, 24, K, S,
                   L.stream()) == L.ERR_UNSUPPORTED


def test_fp16_block_buffer_kernels(L):
    """Config 5 with fp16 block buffers: the *_h / *_h16 entry points read and write fp16 matrices and are otherwise the
    fp32-buffer kernels - checked against those on the same (fp16-representable) values."""
    H = torch.float16
    g = torch.Generator().manual_seed(5)
    st = L.stream()
    # stem: fp16 pooled map == the fp32 one rounded
    n, O = 5, 64
    x = torch.rand(n, 3, 128, 128, generator=g).to(DEV)
    W0 = (torch.randn(O, 3, 7, 7, generator=g) * 0.1).to(DEV)
    sc0, sh0 = (torch.rand(O, generator=g) + 0.5).to(DEV), (torch.randn(O, generator=g) * 0.2).to(DEV)
    o32 = torch.empty(n * 1024, 96, device=DEV)
    o16 = torch.full((n * 1024, 96), 7.0, device=DEV, dtype=H)
    L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(x), L.ptr(W0), L.ptr(o32), 96, n, 3, 128, 128, O, 7, 7, 2, 3, L.ptr(sc0),
           L.ptr(sh0), st)
    L.call('gnx_conv_stem_bnrelu_maxpool_h16', L.ptr(x), L.ptr(W0), L.ptr(o16, H), 96, n, 3, 128, 128, O, 7, 7, 2, 3,
           L.ptr(sc0), L.ptr(sh0), st)
    assert torch.equal(o16[:, :O], o32[:, :O].half()) and float(o16[:, O:].float().min()) == 7.0
    # the same at config 5's real patch size: 256 px (conv map 128 wide, pooled map 64 x 64)
    n2 = 3
    x2 = torch.rand(n2, 3, 256, 256, generator=g).to(DEV)
    p32 = torch.empty(n2 * 4096, 96, device=DEV)
    p16 = torch.full((n2 * 4096, 96), 7.0, device=DEV, dtype=H)
    L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(x2), L.ptr(W0), L.ptr(p32), 96, n2, 3, 256, 256, O, 7, 7, 2, 3, L.ptr(sc0),
           L.ptr(sh0), st)
    L.call('gnx_conv_stem_bnrelu_maxpool_h16', L.ptr(x2), L.ptr(W0), L.ptr(p16, H), 96, n2, 3, 256, 256, O, 7, 7, 2, 3,
           L.ptr(sc0), L.ptr(sh0), st)
    assert torch.equal(p16[:, :O], p32[:, :O].half()) and float(p16[:, O:].float().min()) == 7.0
    ref2 = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(x2.cpu(), W0.cpu(), stride=2, padding=3)
                                                     * sc0.cpu().view(1, -1, 1, 1) + sh0.cpu().view(1, -1, 1, 1)), 3, 2, 1)
    close(p32[:, :O].reshape(n2, 64, 64, O).permute(0, 3, 1, 2), ref2, rtol=2e-4)
    # conv1 on fp16 activations == the fp32-input kernel on the same values
    M, K, N, ct = 2048, 96, 128, 160
    A16 = torch.randn(M, ct, generator=g).half().to(DEV)
    A32 = A16.float()
    W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(DEV), (torch.randn(K, generator=g) * 0.5).to(DEV)
    osc, osh = (torch.rand(N, generator=g) + 0.5).to(DEV), (torch.randn(N, generator=g) * 0.5).to(DEV)
    b1 = torch.empty(M, N, device=DEV, dtype=H)
    b2 = torch.empty(M, N, device=DEV, dtype=H)
    L.call('gnx_conv1x1_bnrelu_f16_act16', L.ptr(A32), ct, L.ptr(W), L.ptr(b1, H), N, M, N, K, L.ptr(sc), L.ptr(sh), L.ptr(osc),
           L.ptr(osh), st)
    L.call('gnx_conv1x1_bnrelu_f16_h', L.ptr(A16, H), ct, L.ptr(W), L.ptr(b2, H), N, M, N, K, L.ptr(sc), L.ptr(sh), L.ptr(osc),
           L.ptr(osh), 0, 0, st)
    assert torch.equal(b1, b2)
    b3 = torch.full((M, N + 8), 7.0, device=DEV, dtype=H)
    W16 = W.half()
    L.call('gnx_conv1x1_bnrelu_h16', L.ptr(A16, H), ct, L.ptr(W16, H), L.ptr(b3, H), N + 8, M, N, K, L.ptr(sc), L.ptr(sh),
           L.ptr(osc), L.ptr(osh), st)                             # the 64-channel-chunk form with fp16 weights: same bits
    assert torch.equal(b3[:, :N], b2) and float(b3[:, N:].float().min()) == 7.0
    # transition (pool, no consumer activation) against torch on the same fp16 values
    nI, S, Kt, Nt = 3, 8, 64, 40
    xt = torch.randn(nI, Kt, S, S, generator=g).half()
    Wt = torch.randn(Nt, Kt, generator=g) * 0.1
    sct, sht = torch.rand(Kt, generator=g) + 0.5, torch.randn(Kt, generator=g) * 0.5
    act = torch.relu(xt.float() * sct.view(1, -1, 1, 1) + sht.view(1, -1, 1, 1))
    ref = F.conv2d(F.avg_pool2d(act, 2, 2).half().double(), Wt.half().double().view(Nt, Kt, 1, 1))
    At = xt.permute(0, 2, 3, 1).reshape(-1, Kt).contiguous().to(DEV)
    So = S // 2
    ot = torch.empty(nI * So * So, Nt, device=DEV, dtype=H)
    Wtd, sctd, shtd = Wt.to(DEV), sct.to(DEV), sht.to(DEV)
    L.call('gnx_conv1x1_bnrelu_f16_h', L.ptr(At, H), Kt, L.ptr(Wtd), L.ptr(ot, H), Nt, nI * So * So, Nt, Kt, L.ptr(sctd),
           L.ptr(shtd), None, None, 1, S, st)
    close(ot.reshape(nI, So, So, Nt).permute(0, 3, 1, 2).double(), ref, rtol=2e-3)
    # the two-step transition (round 2): pooled activation as its own pass, then conv1x1_h16 without prologue / consumer
    # activation - the same fp16 operands, the same k order: the one-kernel form's bits
    Pt = torch.full((nI * So * So, Kt + 8), 7.0, device=DEV, dtype=H)
    L.call('gnx_bnrelu_avgpool2_h16', L.ptr(At, H), Kt, L.ptr(Pt, H), Kt + 8, nI, Kt, S, L.ptr(sctd), L.ptr(shtd), st)
    want_p = F.avg_pool2d(act, 2, 2).permute(0, 2, 3, 1).reshape(-1, Kt)
    close(Pt[:, :Kt].float(), want_p, rtol=1e-3, atol=1e-3)
    assert float(Pt[:, Kt:].float().min()) == 7.0
    ot2 = torch.empty(nI * So * So, Nt, device=DEV, dtype=H)
    L.call('gnx_conv1x1_bnrelu_h16', L.ptr(Pt, H), Kt + 8, L.ptr(Wtd.half(), H), L.ptr(ot2, H), Nt, nI * So * So, Nt, Kt, None,
           None, None, None, st)
    assert torch.equal(ot2, ot)
    # conv2 with an fp16 output == the fp32-output kernel rounded
    S2, n2 = 8, 4
    a16 = torch.relu(torch.randn(n2 * S2 * S2, 128, generator=g)).half().to(DEV)
    Wc = (torch.randn(32, 128, 3, 3, generator=g) * 0.05).to(DEV)
    Wr = torch.empty(9, 32, 128, device=DEV)
    L.call('gnx_repack_conv3x3', L.ptr(Wc), L.ptr(Wr), 32, 128, st)
    Wr16 = Wr.half()
    c32 = torch.empty(n2 * S2 * S2, 40, device=DEV)
    c16 = torch.full((n2 * S2 * S2, 40), 7.0, device=DEV, dtype=H)
    L.call('gnx_conv3x3_f16_dma', L.ptr(a16, H), 128, L.ptr(Wr16, H), L.ptr(c32), 40, n2 * S2 * S2, 32, 128, S2, st)
    L.call('gnx_conv3x3_f16_dma_h', L.ptr(a16, H), 128, L.ptr(Wr16, H), L.ptr(c16, H), 40, n2 * S2 * S2, 32, 128, S2, st)
    assert torch.equal(c16[:, :32], c32[:, :32].half()) and float(c16[:, 32:].float().min()) == 7.0
    # final pool reading fp16
    f16 = torch.randn(6 * 16, 200, generator=g).half().to(DEV)
    scf, shf = (torch.rand(200, generator=g) + 0.5).to(DEV), (torch.randn(200, generator=g) * 0.3).to(DEV)
    p32, p16 = torch.empty(6, 200, device=DEV), torch.empty(6, 200, device=DEV)
    f32 = f16.float()
    L.call('gnx_bnrelu_avgpool', L.ptr(f32), 200, L.ptr(p32), 200, 6, 200, 16, L.ptr(scf), L.ptr(shf), st)
    L.call('gnx_bnrelu_avgpool_h16', L.ptr(f16, H), 200, L.ptr(p16), 200, 6, 200, 16, L.ptr(scf), L.ptr(shf), st)
    assert torch.equal(p32, p16)


def _dense_layer_ref(x16, W1, W2, sc1, sh1, sc2, sh2):
    """densenet.py:35-44 on fp16 values with the fused kernel's rounding points (operands fp16, sums exact in double):
    x16 [n, S, S, K] halves -> [n, S, S, 32] double."""
    a = torch.relu(torch.addcmul(sh1, x16.float(), sc1)).half()                   # norm1 -> relu1, rounded once
    y = torch.einsum('nyxk,ok->nyxo', a.double(), W1.half().double())             # conv1 (1x1)
    b = torch.relu(y.float() * sc2 + sh2).half()                                  # norm2 -> relu2, the fp16 bottleneck
    o = F.conv2d(b.double().permute(0, 3, 1, 2), W2.half().double(), padding=1)   # conv2 (3x3, pad 1)
    return o.permute(0, 2, 3, 1)


@pytest.mark.parametrize("S,n,K,N", [(64, 2, 256, 128), (32, 3, 512, 256), (16, 6, 1024, 512), (8, 8, 64, 128),
                                     (16, 1040, 96, 128), (32, 5, 160, 384)])
def test_transition_f16_fused(L, S, n, K, N):
    """gnx_transition_f16 (norm -> relu -> 1x1 conv -> avgpool 2x2 as one kernel, evaluated pool-first on channel-blocked fp16
    buffers) against the transition evaluated in double on the same fp16 values with the kernel's rounding points: the pooled
    activated operand rounded to fp16 (fp32 arithmetic in the order of gnx_bnrelu_avgpool2_h16), fp16 weights, exact sums,
    fp16 output.  Shapes: every map size, one and two channel blocks per wave, two output passes, more steps than CUs."""
    g = torch.Generator().manual_seed(S * 1000 + K + n)
    x = torch.randn(n, S, S, K, generator=g).half()
    W = torch.randn(N, K, generator=g) * (1.0 / K ** 0.5)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    sc[::5] *= -1.0
    H = torch.float16
    st = L.stream()
    X = x.reshape(-1, K // 32, 32).permute(1, 0, 2).contiguous().to(DEV)       # channel-blocked [K / 32][rows][32]
    so = S // 2
    rows_out = n * so * so
    cy = N + 64                                                                 # the output buffer is wider than the N columns
    Y = torch.full((cy // 32, rows_out, 32), 7.0, dtype=H, device=DEV)
    wp = torch.empty(N * K, device=DEV, dtype=H)
    Wd, scd, shd = W.to(DEV), sc.to(DEV), sh.to(DEV)
    L.call('gnx_transition_f16_pack', L.ptr(Wd), L.ptr(wp, H), N, K, st)
    L.call('gnx_transition_f16', L.ptr(X, H), X.shape[1], n, S, K, N, L.ptr(wp, H), L.ptr(scd), L.ptr(shd), L.ptr(Y, H), rows_out, st)
    torch.cuda.synchronize()
    got = Y.cpu().permute(1, 0, 2).reshape(n, so, so, cy)
    assert float(got[..., N:].float().min()) == 7.0 and float(got[..., N:].float().max()) == 7.0
    a = torch.relu(torch.addcmul(sh, x.float(), sc))                            # fp32, as the kernel
    p = (0.25 * (((a[:, 0::2, 0::2] + a[:, 0::2, 1::2]) + a[:, 1::2, 0::2]) + a[:, 1::2, 1::2])).half()
    ref = torch.einsum('nyxk,ok->nyxo', p.double(), W.half().double())
    out = got[..., :N].double()
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max().item()
    tol = 2e-3 * ref.abs().max().item()
    assert err <= tol, "S=%d n=%d K=%d N=%d: max abs err %.3e > %.3e" % (S, n, K, N, err, tol)
    # the TAPED form (gnx_transition_f16_tape, the forward of the fp16 gradient path): the same output bit for bit, plus the
    # pooled activated operand as a row-major [pooled pixel][ldp] matrix - bit for bit the pooling pass's output (the fp32
    # arithmetic above in the same order, one rounding) - and nothing beyond its K columns or its rows
    Yt = torch.full((cy // 32, rows_out, 32), 7.0, dtype=H, device=DEV)
    ldp = K + 8
    Pt = torch.full((rows_out + 3, ldp), 5.0, dtype=H, device=DEV)
    L.call('gnx_transition_f16_tape', L.ptr(X, H), X.shape[1], n, S, K, N, L.ptr(wp, H), L.ptr(scd), L.ptr(shd), L.ptr(Yt, H), rows_out,
           L.ptr(Pt, H), ldp, st)
    torch.cuda.synchronize()
    assert torch.equal(Yt, Y), "the taped form changed the transition's output"
    assert torch.equal(Pt[:rows_out, :K].cpu().reshape(n, so, so, K), p), "taped pooled operand"
    assert float(Pt[:rows_out, K:].float().min()) == 5.0 and float(Pt[rows_out:].float().min()) == 5.0
    pool2 = torch.empty((rows_out, K), dtype=H, device=DEV)
    L.call('gnx_bnrelu_avgpool2_h16_cb', L.ptr(X, H), X.shape[1], L.ptr(pool2, H), K, n, K, S, L.ptr(scd), L.ptr(shd), st)
    torch.cuda.synchronize()
    assert torch.equal(Pt[:rows_out, :K], pool2), "the tape differs from the pooling pass it replaces"


@pytest.mark.parametrize("S,n,K,ct", [(64, 2, 64, 128), (64, 3, 96, 160), (64, 260, 64, 128), (64, 5, 224, 288),
                                      (64, 258, 160, 224), (64, 2, 480, 544), (32, 9, 992, 1056), (32, 4, 128, 192),
                                      (32, 300, 224, 288), (16, 8, 256, 320), (16, 520, 96, 160), (8, 16, 512, 576),
                                      (8, 1040, 64, 128), (4, 64, 992, 1056), (4, 4160, 64, 128),
                                      (64, 530, 96, 160), (32, 1100, 160, 224), (32, 515, 480, 544)])
@pytest.mark.parametrize("form", [0, 1])
def test_dense_layer_f16_fused(L, S, n, K, ct, form):
    """(form 1: the k-split kernel of 64 x 64 / 32 x 32 maps with K <= 512, gnx_dense_layer_f16_set_form; other shapes run the
    same kernel under both forms.)  gnx_dense_layer_f16 (one kernel per dense layer, bottleneck in LDS only) against the layer evaluated in double on the
    same fp16 values with the same rounding points.  Tolerance 3e-3 of the output range: the fp16 rounding of the output
    (2^-11 relative) plus bottleneck values that round the other way when conv1's fp32 sum runs in another order.  Also: the
    input columns and everything beyond the 32 new columns stay untouched.  Shapes: every map size the kernel takes, K from
    2 to 31 stages, single-step and swept images, fewer and more work units than compute units (260 ... 4160 images)."""
    if form == 1 and not (S >= 32 and K <= 512):
        pytest.skip("the k-split form takes 64 x 64 and 32 x 32 maps with K <= 512")
    L.call('gnx_dense_layer_f16_set_form', form)
    try:
        _dense_layer_f16_fused_case(L, S, n, K, ct)
    finally:
        L.call('gnx_dense_layer_f16_set_form', 0)


def _dense_layer_f16_fused_case(L, S, n, K, ct):
    g = torch.Generator().manual_seed(S * 1000 + K + n)
    x = torch.randn(n, S, S, ct, generator=g).half()
    x[:, :, :, K:] = 7.0
    W1 = torch.randn(128, K, generator=g) * (1.0 / K ** 0.5)
    W2 = torch.randn(32, 128, 3, 3, generator=g) * 0.05
    sc1, sh1 = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    sc1[::7] *= -1.0                                                               # negative scales too
    sc2, sh2 = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.5
    st = L.stream()
    H = torch.float16
    # the kernel's buffer is channel-blocked: [ct / 32][rows][32]
    def blocked(t):
        return t.reshape(-1, ct // 32, 32).permute(1, 0, 2).contiguous()

    def rowmajor(b):
        return b.permute(1, 0, 2).reshape(-1, ct)
    X = blocked(x).to(DEV)
    rows = X.shape[1]
    W1d, W2d = W1.to(DEV), W2.to(DEV)
    w1p = torch.empty(128 * K, device=DEV, dtype=H)
    w2p = torch.empty(9 * 8 * 512, device=DEV, dtype=H)
    L.call('gnx_dense_layer_f16_pack', L.ptr(W1d), L.ptr(W2d), L.ptr(w1p, H), L.ptr(w2p, H), K, st)
    d = [v.to(DEV) for v in (sc1, sh1, sc2, sh2)]
    L.call('gnx_dense_layer_f16', L.ptr(X, H), rows, n, S, K, L.ptr(w1p, H), L.ptr(w2p, H), L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]),
           L.ptr(d[3]), st)
    torch.cuda.synchronize()
    got = rowmajor(X.cpu()).reshape(n, S, S, ct)
    assert torch.equal(got[..., :K], x[..., :K]), "input columns changed"
    assert float(got[..., K + 32:].float().min()) == 7.0 and float(got[..., K + 32:].float().max()) == 7.0
    pick = sorted(set(i for i in (0, 1, 2, 3, 255, 256, 257, 511, 512, n // 2, n - 2, n - 1) if 0 <= i < n))
    ref = _dense_layer_ref(x[pick][..., :K], W1, W2, sc1, sh1, sc2, sh2)
    out = got[pick][..., K:K + 32].double()
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max().item()
    tol = 3e-3 * ref.abs().max().item()
    assert err <= tol, "S=%d n=%d K=%d: max abs err %.3e > %.3e" % (S, n, K, err, tol)
    # the TAPED form (gnx_dense_layer_f16_tape, the forward of the fp16 gradient path): the same output bit for bit, plus the
    # activated bottleneck copied out of the LDS tile as [4][rows][32] halves - against relu2(norm2(conv1(relu1(norm1(x)))))
    # in double with the kernel's rounding points (one fp16 rounding of the value; conv1's fp32 sum in another order can
    # move a value by one more ulp: 2e-3 of the range)
    Xt = blocked(x).to(DEV)
    At = torch.full((4, rows + 5, 32), 3.0, device=DEV, dtype=H)
    L.call('gnx_dense_layer_f16_tape', L.ptr(Xt, H), rows, n, S, K, L.ptr(w1p, H), L.ptr(w2p, H), L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]),
           L.ptr(d[3]), L.ptr(At, H), rows + 5, st)
    torch.cuda.synchronize()
    assert torch.equal(Xt, X), "the taped form changed the layer's output"
    assert float(At[:, rows:].float().min()) == 3.0 and float(At[:, rows:].float().max()) == 3.0
    a_got = At[:, :rows].permute(1, 0, 2).reshape(n, S, S, 128).cpu()[pick].double()
    xa = torch.relu(torch.addcmul(sh1, x[pick][..., :K].float(), sc1)).half()
    a_ref = torch.relu(torch.einsum('nyxk,ok->nyxo', xa.double(), W1.half().double()).float() * sc2 + sh2).half().double()
    a_err = (a_got - a_ref).abs().max().item()
    assert a_err <= 2e-3 * a_ref.abs().max().item(), "S=%d n=%d K=%d: taped bottleneck max abs err %.3e" % (S, n, K, a_err)
    # every image, cheaply: the column sums of the new channels against the same sums of a second launch on a permuted
    # batch (a work unit is an image or a tile of whole images: the result must not depend on which workgroup ran it)
    if n >= 16:
        perm = torch.randperm(n, generator=g)
        X2 = blocked(x[perm]).to(DEV)
        L.call('gnx_dense_layer_f16', L.ptr(X2, H), rows, n, S, K, L.ptr(w1p, H), L.ptr(w2p, H), L.ptr(d[0]), L.ptr(d[1]),
               L.ptr(d[2]), L.ptr(d[3]), st)
        torch.cuda.synchronize()
        assert torch.equal(rowmajor(X2.cpu()).reshape(n, S, S, ct), got[perm]), "result depends on the image's position in the batch"


@pytest.mark.parametrize("n,O,P", [(3, 64, 128), (300, 64, 128), (2, 32, 128), (3, 64, 256), (270, 64, 256), (2, 32, 256)])
def test_stem_fused_with_norm0_relu0_pool0(L, n, O, P):
    """conv0 -> norm0 -> relu0 -> pool0 in one kernel (128- and 256-px geometry) vs torch; other geometries must say
    UNSUPPORTED."""
    g = torch.Generator().manual_seed(n + O)
    x = torch.rand(n, 3, P, P, generator=g)
    W = torch.randn(O, 3, 7, 7, generator=g) * 0.1
    sc, sh = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2
    ref = F.max_pool2d(torch.relu(F.conv2d(x, W, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)),
                       3, 2, 1)
    xd, Wd, scd, shd = x.to(DEV), W.to(DEV), sc.to(DEV), sh.to(DEV)
    Q = P // 4
    out = torch.full((n * Q * Q, O + 8), 7.0, device=DEV)
    L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(xd), L.ptr(Wd), L.ptr(out), O + 8, n, 3, P, P, O, 7, 7, 2, 3,
           L.ptr(scd), L.ptr(shd), L.stream())
    close(out[:, :O].reshape(n, Q, Q, O).permute(0, 3, 1, 2), ref, rtol=1e-4)
    assert float(out[:, O:].min()) == 7.0
    x64 = torch.rand(2, 3, 64, 64, device=DEV)
    assert L.query('gnx_conv_stem_bnrelu_maxpool', L.ptr(x64), L.ptr(Wd), L.ptr(out), O + 8, 2, 3, 64, 64, O, 7, 7, 2, 3,
                   L.ptr(scd), L.ptr(shd), L.stream()) == L.ERR_UNSUPPORTED


@pytest.mark.parametrize("n,S,K", [(16, 4, 32), (4, 8, 128), (8, 16, 64), (1, 32, 128), (40, 32, 128), (1, 64, 32),
                                   (4160, 4, 128), (5, 8, 64), (3, 4, 32), (1001, 8, 32)])      # last three: ragged last tile
def test_conv3x3_winograd(L, n, S, K):
    """Winograd F(2,3)-along-x form of conv2 on a pre-activated operand == the direct 3x3 convolution (rounding only)."""
    N = 32
    g = torch.Generator().manual_seed(S * 10 + K)
    x = torch.randn(n, K, S, S, generator=g)
    W = torch.randn(N, K, 3, 3, generator=g) * 0.1
    ref = F.conv2d(x.double(), W.double(), padding=1).float()
    A = x.permute(0, 2, 3, 1).reshape(-1, K).contiguous().to(DEV)
    Wd = W.to(DEV)
    Wu = torch.empty((12, N, K), device=DEV)
    L.call('gnx_winograd_conv3x3_weights', L.ptr(Wd), L.ptr(Wu), N, K, L.stream())
    g0, g1, g2 = W[..., 0], W[..., 1], W[..., 2]                         # [N][K][ky]
    Uref = torch.stack([g0, 0.5 * ((g0 + g2) + g1), 0.5 * ((g0 + g2) - g1), g2], 0)     # [xi][N][K][ky]
    close(Wu.view(3, 4, N, K), Uref.permute(3, 0, 1, 2), rtol=1e-6, what='transformed weights')
    ldc = N + 4
    out = torch.full((n * S * S, ldc), 7.0, device=DEV)
    L.call('gnx_conv3x3_winograd', L.ptr(A), K, L.ptr(Wu), out.data_ptr() + 4 * 2, ldc, n * S * S, N, K, S, L.stream())
    close(out[:, 2:2 + N].reshape(n, S, S, N).permute(0, 3, 1, 2), ref, rtol=2e-4)
    assert float(out[:, :2].min()) == 7.0 and float(out[:, 2 + N:].min()) == 7.0
    assert L.query('gnx_conv3x3_winograd', L.ptr(A), K, L.ptr(Wu), L.ptr(out), ldc, n * S * S, 16, K, S,
                   L.stream()) == L.ERR_UNSUPPORTED


@pytest.mark.parametrize("n,P,O,KH,stride,pad", [(3, 32, 8, 7, 2, 3), (2, 128, 64, 7, 2, 3), (2, 16, 10, 3, 1, 1),
                                                 (1, 30, 24, 7, 2, 3)])
def test_stem_conv_and_pools(L, n, P, O, KH, stride, pad):
    g = torch.Generator().manual_seed(P + O)
    x = torch.rand(n, 3, P, P, generator=g)
    W = torch.randn(O, 3, KH, KH, generator=g) * 0.1
    ref = F.conv2d(x, W, stride=stride, padding=pad)
    Ho = ref.shape[2]
    out = torch.zeros((n * Ho * Ho, O), device=DEV)
    xd, Wd = x.to(DEV), W.to(DEV)
    L.call('gnx_conv_stem', L.ptr(xd), L.ptr(Wd), L.ptr(out), O, n, 3, P, P, O, KH, KH, stride, pad,
           L.stream())
    close(out.reshape(n, Ho, Ho, O).permute(0, 3, 1, 2), ref, rtol=1e-4)
    sc, sh = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2
    act = torch.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    mp = F.max_pool2d(act, 3, 2, 1)
    Hp = mp.shape[2]
    pooled = torch.zeros((n * Hp * Hp, O + 4), device=DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    L.call('gnx_bnrelu_maxpool', L.ptr(out), O, L.ptr(pooled), O + 4, n, O, Ho, Ho, L.ptr(scd),
           L.ptr(shd), L.stream())
    close(pooled[:, :O].reshape(n, Hp, Hp, O).permute(0, 3, 1, 2), mp, rtol=1e-4)
    gap = torch.zeros((n, O), device=DEV)
    L.call('gnx_bnrelu_avgpool', L.ptr(out), O, L.ptr(gap), O, n, O, Ho * Ho, L.ptr(scd), L.ptr(shd),
           L.stream())
    close(gap, act.mean((2, 3)), rtol=1e-4)


# ----------------------------------------------------------------------------------------------- round 2 additions
@pytest.mark.parametrize("C,Hi", [(8, 12), (64, 32), (6, 9)])
def test_maxpool_backward_routes_ties_like_torch(L, C, Hi):
    """pool0's adjoint on maps with TIES (piecewise-constant conv0 maps: white slide background, all-zero background spots):
    torch sends a window's gradient to the first maximal element of its row-major scan, not to every maximal one.
    Found by the C2-shape gradient test (closed-form images, norm0.weight off by 8x its magnitude)."""
    g = torch.Generator().manual_seed(C + Hi)
    n = 3
    # a conv0 map that is constant on 3x2 blocks plus a few random cells, channels-last rows
    coarse = torch.randint(-2, 4, (n, C, (Hi + 2) // 3, (Hi + 1) // 2), generator=g).float()
    raw = coarse.repeat_interleave(3, 2).repeat_interleave(2, 3)[:, :, :Hi, :Hi].contiguous()
    raw[:, :, 1::5, 2::7] += torch.rand(raw[:, :, 1::5, 2::7].shape, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    xr = raw.clone().requires_grad_(True)
    act = torch.relu(xr * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    act.retain_grad()
    mp = F.max_pool2d(act, 3, 2, 1)
    Ho = mp.shape[2]
    dout = torch.randn(mp.shape, generator=g)
    mp.backward(dout)
    rows = raw.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    pooled = torch.empty((n * Ho * Ho, C), device=DEV)
    L.call('gnx_bnrelu_maxpool', L.ptr(rows), C, L.ptr(pooled), C, n, C, Hi, Hi, L.ptr(scd), L.ptr(shd), L.stream())
    close(pooled.reshape(n, Ho, Ho, C).permute(0, 3, 1, 2), mp.detach(), rtol=1e-6, what='pooled')
    dO = dout.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV)
    dAct = torch.empty((n * Hi * Hi, C), device=DEV)
    L.call('gnx_maxpool_bwd', L.ptr(rows), C, L.ptr(pooled), C, L.ptr(dO), C, L.ptr(dAct), C, n, C, Hi, Hi, L.ptr(scd),
           L.ptr(shd), L.stream())
    # torch's d/d(act) includes elements at exactly 0 (a window of zeros): the ReLU mask removes them either way
    want = (act.grad * (act.detach() > 0)).permute(0, 2, 3, 1).reshape(-1, C)
    close(dAct, want, rtol=1e-6, atol=1e-7, what='dAct with ties')
    # the training path's form: the forward records the winning window element, the adjoint routes by index - this is
    # torch's d/d(act) itself, zeros included (the BN adjoint that follows applies the ReLU mask)
    pooled2 = torch.empty_like(pooled)
    amax = torch.full((n * Ho * Ho, C), 99, device=DEV, dtype=torch.uint8)
    L.call('gnx_bnrelu_maxpool_argmax', L.ptr(rows), C, L.ptr(pooled2), C, amax.data_ptr(), n, C, Hi, Hi, L.ptr(scd), L.ptr(shd),
           L.stream())
    assert torch.equal(pooled2, pooled) and int(amax.max()) <= 8
    if C % 4 == 0:
        dAct2 = torch.empty_like(dAct)
        L.call('gnx_maxpool_bwd_argmax', amax.data_ptr(), L.ptr(dO), C, L.ptr(dAct2), C, n, C, Hi, Hi, L.stream())
        close(dAct2, act.grad.permute(0, 2, 3, 1).reshape(-1, C), rtol=1e-6, atol=1e-7, what='dAct by index')


@pytest.mark.parametrize("M,K,ct", [(512, 992, 1024), (2048, 480, 1024), (8192, 256, 512), (640, 352, 512)])
def test_conv1x1_small_batch_split_k(L, M, K, ct):
    """gnx_conv1x1_bnrelu_ws: conv1 of a dense layer on the matrices a batch of 32 patches gives (512 - 8192 rows): K split
    over workgroups, partial tiles summed in a fixed order == the unsplit entry point up to summation order; fp64 reference."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, ct, generator=g).to(DEV)
    W = (torch.randn(128, K, generator=g) * 0.05).to(DEV)
    sc, sh = (torch.rand(K, generator=g) + 0.5).to(DEV), (torch.randn(K, generator=g) * 0.1).to(DEV)
    nws = L.query('gnx_conv1x1_workspace', M, 128, K)
    assert nws > 0
    ws = torch.empty(nws, device=DEV)
    out1, out2, out0 = (torch.empty(M, 128, device=DEV) for _ in range(3))
    st = L.stream()
    L.call('gnx_conv1x1_bnrelu_ws', L.ptr(A), ct, L.ptr(W), L.ptr(out1), 128, M, 128, K, L.ptr(sc), L.ptr(sh), L.ptr(ws), st)
    L.call('gnx_conv1x1_bnrelu_ws', L.ptr(A), ct, L.ptr(W), L.ptr(out2), 128, M, 128, K, L.ptr(sc), L.ptr(sh), L.ptr(ws), st)
    L.call('gnx_conv1x1_bnrelu', L.ptr(A), ct, L.ptr(W), L.ptr(out0), 128, M, 128, K, L.ptr(sc), L.ptr(sh), 0, 0, st)
    assert torch.equal(out1, out2)
    ref = torch.relu(A[:, :K].double() * sc.double() + sh.double()) @ W.double().t()
    scale = ref.abs().max().item()
    assert (out1.double() - ref).abs().max().item() <= 2e-6 * K ** 0.5 * scale
    assert (out0.double() - ref).abs().max().item() <= 2e-6 * K ** 0.5 * scale


@pytest.mark.parametrize("P", [128, 256])
def test_fused_stem_records_pool0_winner_like_torch(L, P):
    """gnx_conv_stem_bnrelu_maxpool_argmax (the f-trained step's stem under running statistics): pooled output bit-equal to
    the plain fused stem, recorded window index == torch's max_pool2d indices on the kernel's own activated map where the
    maximum is unique, and the first maximal element of the scan on patches with TIES (constant and all-zero patches: every
    window element equal); then the index adjoint + norm0/relu0 (gnx_maxpool_bwd_argmax_bnrelu) against autograd."""
    g = torch.Generator().manual_seed(P)
    n, O = 4, 64
    x = torch.rand(n, 3, P, P, generator=g)
    x[1] = 0.0                                            # an all-zero background spot: conv0 map == 0 everywhere
    x[2] = 0.5                                            # a constant patch: conv0 map constant away from the border
    w = torch.randn(O, 3, 7, 7, generator=g) * 0.1
    sc, sh = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2 + 0.1
    xd, wd, scd, shd = x.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV)
    hs, hp = P // 2, P // 4
    pooled = torch.empty((n * hp * hp, O), device=DEV)
    pooled2 = torch.empty_like(pooled)
    amax = torch.full((n * hp * hp, O), 99, device=DEV, dtype=torch.uint8)
    st = L.stream()
    L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(xd), L.ptr(wd), L.ptr(pooled), O, n, 3, P, P, O, 7, 7, 2, 3, L.ptr(scd),
           L.ptr(shd), st)
    L.call('gnx_conv_stem_bnrelu_maxpool_argmax', L.ptr(xd), L.ptr(wd), L.ptr(pooled2), O, amax.data_ptr(), n, 3, P, P, O,
           7, 7, 2, 3, L.ptr(scd), L.ptr(shd), st)
    assert torch.equal(pooled, pooled2) and int(amax.max()) <= 8
    # the kernel's own conv0 map is not observable; torch's conv on the device gives the same map up to rounding
    pre = F.conv2d(xd, wd, stride=2, padding=3)
    act = torch.relu(pre * scd.view(1, -1, 1, 1) + shd.view(1, -1, 1, 1))
    mp, ind = F.max_pool2d(act, 3, 2, 1, return_indices=True)
    close(pooled.reshape(n, hp, hp, O).permute(0, 3, 1, 2), mp, rtol=1e-5, atol=1e-5, what='pooled')
    # window index -> flat index of the map
    am = amax.reshape(n, hp, hp, O).permute(0, 3, 1, 2).long()
    oy = torch.arange(hp, device=DEV).view(1, 1, hp, 1)
    ox = torch.arange(hp, device=DEV).view(1, 1, 1, hp)
    iy, ix = 2 * oy + am // 3 - 1, 2 * ox + am % 3 - 1
    assert bool(((iy >= 0) & (iy < hs) & (ix >= 0) & (ix < hs)).all())          # never a padding element
    mine = iy * hs + ix
    # exact ties (patches 1 and 2, interior): torch's first maximum; elsewhere: equal unless the top two are within rounding
    assert torch.equal(mine[1], ind[1])
    inner = (slice(None), slice(2, hp - 2), slice(2, hp - 2))
    assert int(am[2][inner].max()) == 0                  # constant interior (torch's own conv may not be bit-constant there)
    picked = act.flatten(2).gather(2, mine.flatten(2)).reshape(mp.shape)
    close(picked, mp, rtol=1e-5, atol=1e-5, what='value at the recorded index')
    assert (mine[[0, 3]] != ind[[0, 3]]).float().mean().item() < 1e-4      # random patches: unique maxima but for rounding
    # adjoint: d(conv0 map) by index + norm0/relu0 == autograd through relu(bn(.)) -> max_pool2d at the recorded winners
    dO = torch.randn(n * hp * hp, O, generator=g).to(DEV)
    dPre = torch.empty((n * hs * hs, O), device=DEV)
    L.call('gnx_maxpool_bwd_argmax_bnrelu', amax.data_ptr(), L.ptr(dO), O, L.ptr(pooled), O, L.ptr(scd), L.ptr(dPre), O, n, O,
           hs, hs, st)
    want = torch.zeros(n, O, hs * hs, device=DEV)
    dOm = dO.reshape(n, hp * hp, O).permute(0, 2, 1) * (pooled.reshape(n, hp * hp, O).permute(0, 2, 1) > 0)
    want.scatter_add_(2, mine.flatten(2), dOm)
    want = (want * scd.view(1, -1, 1)).permute(0, 2, 1).reshape(-1, O)
    close(dPre, want, rtol=1e-6, atol=1e-7, what='d conv0 map')


def test_uint8_patches_equal_float_patches_bit_for_bit(L):
    """patches kept as uint8 up to the stem's operand load.  ToTensor's u8 / 255 (and Normalize's (v - m) / s)
    inside the kernels must be the floats torch computes on the host: all 256 byte values x 3 channels through
    gnx_u8_to_f32, then the fused u8 stem against the fused float stem on the torch-converted patches - bit for bit."""
    st = L.stream()
    # every byte value in every channel
    x8 = torch.arange(256, dtype=torch.uint8).repeat(3, 4).reshape(1, 3, 4, 256).contiguous()
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    nrm = torch.cat([mean, std, 1.0 / std]).to(DEV)
    out = torch.empty((1, 3, 4, 256), device=DEV)
    L.call('gnx_u8_to_f32', x8.to(DEV).data_ptr(), L.ptr(out), 1, 3, 4, 256, None, st)
    assert torch.equal(out.cpu(), x8.float().div(255))                          # torchvision ToTensor
    L.call('gnx_u8_to_f32', x8.to(DEV).data_ptr(), L.ptr(out), 1, 3, 4, 256, L.ptr(nrm), st)
    want = x8.float().div(255).sub(mean.view(1, 3, 1, 1)).div(std.view(1, 3, 1, 1))    # + Normalize
    assert torch.equal(out.cpu(), want)
    g = torch.Generator().manual_seed(8)
    O = 64
    W0 = (torch.randn(O, 3, 7, 7, generator=g) * 0.1).to(DEV)
    sc0, sh0 = (torch.rand(O, generator=g) + 0.5).to(DEV), (torch.randn(O, generator=g) * 0.2).to(DEV)
    for P, n in ((128, 5), (256, 3)):
        u8 = torch.randint(0, 256, (n, 3, P, P), generator=g, dtype=torch.uint8)
        u8[0] = 255                                                             # a saturated patch
        u8[-1, :, : P // 2] = 0                                                 # and a half-empty one
        S = P // 4
        for norm in (None, nrm):
            xf = u8.float().div(255)
            if norm is not None:
                xf = xf.sub(mean.view(1, 3, 1, 1)).div(std.view(1, 3, 1, 1))
            o_f = torch.empty(n * S * S, 96, device=DEV)
            o_u = torch.full((n * S * S, 96), 7.0, device=DEV)
            xfd = xf.to(DEV)
            L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(xfd), L.ptr(W0), L.ptr(o_f), 96, n, 3, P, P, O, 7, 7, 2, 3,
                   L.ptr(sc0), L.ptr(sh0), st)
            L.call('gnx_conv_stem_bnrelu_maxpool_u8', u8.to(DEV).data_ptr(), L.ptr(W0), o_u.data_ptr(), 96, n, 3, P, P, O,
                   7, 7, 2, 3, L.ptr(sc0), L.ptr(sh0), L.ptr(norm), 0, st)
            assert torch.equal(o_u[:, :O], o_f[:, :O]) and float(o_u[:, O:].min()) == 7.0
            o_h = torch.empty(n * S * S, 96, device=DEV, dtype=torch.float16)
            L.call('gnx_conv_stem_bnrelu_maxpool_u8', u8.to(DEV).data_ptr(), L.ptr(W0), o_h.data_ptr(), 96, n, 3, P, P, O,
                   7, 7, 2, 3, L.ptr(sc0), L.ptr(sh0), L.ptr(norm), 1, st)
            assert torch.equal(o_h[:, :O], o_f[:, :O].half())
    # geometry the fused stem does not take -> UNSUPPORTED (the caller converts and takes the float stems)
    u8 = torch.zeros((1, 3, 64, 64), dtype=torch.uint8, device=DEV)
    o = torch.empty(256, 64, device=DEV)
    assert L.query('gnx_conv_stem_bnrelu_maxpool_u8', u8.data_ptr(), L.ptr(W0), o.data_ptr(), 64, 1, 3, 64, 64, O, 7, 7, 2, 3,
                   L.ptr(sc0), L.ptr(sh0), None, 0, st) == L.ERR_UNSUPPORTED


@pytest.mark.parametrize("M,N,K,ldx,act", [(64 * 40, 128, 64, 96, True), (64 * 37 + 13, 128, 96, 256, True),
                                           (5000, 128, 416, 512, True), (64 * 20, 256, 128, 128, False),
                                           (300, 128, 992, 1024, True),
                                           # enough tiles for a multiple-of-8 split count: the round-2 128 x 256 kernel
                                           (64 * 300 + 13, 128, 416, 512, True), (40000, 256, 224, 256, True),
                                           (20000, 128, 992, 1024, True), (33000, 128, 160, 160, False)])
def test_wgrad_1x1_transposed_image_kernel(L, M, N, K, ldx, act):
    """gnx_wgrad_bnrelu(taps = 1) on the dense layers' shapes (the round-2 [channel][position] kernel: 128 | N, 4 | K) vs
    fp64: dW[n][k] = sum_m dY[m][n] relu(scale[k] X[m][k] + shift[k]).  Ragged last tile, K not a multiple of the
    workgroup's 128-column range, leading dimensions wider than the operands."""
    g = torch.Generator().manual_seed(M + K)
    X = torch.randn(M, ldx, generator=g)
    dY = torch.randn(M, N + 32, generator=g)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    A = torch.relu(X[:, :K].double() * sc.double() + sh.double()) if act else X[:, :K].double()
    ref = dY[:, :N].double().t() @ A
    Xd, dYd, scd, shd = X.to(DEV), dY.to(DEV), sc.to(DEV), sh.to(DEV)
    dW = torch.full((N, K), 7.0, device=DEV)
    ws = torch.empty(L.query('gnx_wgrad_workspace', M, N, K, 1), device=DEV)
    L.call('gnx_wgrad_bnrelu', L.ptr(dYd), N + 32, L.ptr(Xd), ldx, L.ptr(scd) if act else None, L.ptr(shd) if act else None,
           L.ptr(dW), L.ptr(ws), M, N, K, 0, 1, 0, 0, L.stream())
    close(dW, ref, rtol=2e-5, atol=1e-5, what='dW 1x1')
    L.call('gnx_wgrad_bnrelu', L.ptr(dYd), N + 32, L.ptr(Xd), ldx, L.ptr(scd) if act else None, L.ptr(shd) if act else None,
           L.ptr(dW), L.ptr(ws), M, N, K, 0, 1, 0, 1, L.stream())                    # accumulate
    close(dW, 2 * ref, rtol=2e-5, atol=1e-5, what='dW 1x1 accumulated')


@pytest.mark.parametrize("n,S,K,act", [(3, 32, 128, False), (5, 16, 128, True), (9, 8, 128, True), (4, 8, 256, False),
                                       (2, 4, 128, True), (1, 32, 128, True)])
def test_wgrad_3x3_transposed_image_kernel(L, n, S, K, act):
    """gnx_wgrad_bnrelu(taps = 9) vs torch's conv2d weight gradient in fp64: maps of 8 / 16 / 32 (round-2 kernel: shifted
    dY copies, border rows skipped by scalar branches) and 4 (the r1 kernel); several images per tile and tiles inside
    an image, so that every border case of every tap is hit."""
    g = torch.Generator().manual_seed(n * 100 + S)
    N, M = 32, n * S * S
    ld = 160
    X = torch.randn(M, K, generator=g)
    dYfull = torch.randn(M, ld, generator=g)
    c0 = 64                                                    # the 32 gradient columns sit inside a wider block buffer
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    A = torch.relu(X.double() * sc.double() + sh.double()) if act else X.double()
    x4 = A.reshape(n, S, S, K).permute(0, 3, 1, 2).contiguous().requires_grad_(False)
    w = torch.zeros(N, K, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x4, w, padding=1)
    y.backward(dYfull[:, c0:c0 + N].double().reshape(n, S, S, N).permute(0, 3, 1, 2).contiguous())
    Xd, dYd, scd, shd = X.to(DEV), dYfull.to(DEV), sc.to(DEV), sh.to(DEV)
    dW = torch.full((N, K, 3, 3), 7.0, device=DEV)
    ws = torch.empty(L.query('gnx_wgrad_workspace', M, N, K, 9), device=DEV)
    L.call('gnx_wgrad_bnrelu', dYd.data_ptr() + 4 * c0, ld, L.ptr(Xd), K, L.ptr(scd) if act else None,
           L.ptr(shd) if act else None, L.ptr(dW), L.ptr(ws), M, N, K, S, 9, 0, 0, L.stream())
    close(dW, w.grad, rtol=2e-5, atol=1e-5, what='dW 3x3')


@pytest.mark.parametrize("P,n", [(128, 5), (256, 3)])
def test_stem_with_fp16_matrix_operands(L, P, n):
    """Config 5's stem (gnx_conv_stem_bnrelu_maxpool_f16mul): patch and weights rounded to fp16 at the LDS stash, fp16 MFMA with
    fp32 accumulation, fp16 pooled map - against torch on the SAME rounded operands (fp64 accumulate), from float patches and
    from uint8 patches (with a Normalize)."""
    H16 = torch.float16
    g = torch.Generator().manual_seed(P)
    O = 64
    W0 = torch.randn(O, 3, 7, 7, generator=g) * 0.1
    sc0, sh0 = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.2
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    nrm = torch.cat([mean, std, 1.0 / std]).to(DEV)
    u8 = torch.randint(0, 256, (n, 3, P, P), generator=g, dtype=torch.uint8)
    S = P // 4
    for kind in ('float', 'u8', 'u8norm'):
        if kind == 'float':
            xf = torch.rand(n, 3, P, P, generator=g)
            arg, is_u8, norm = xf.to(DEV), 0, None
        else:
            xf = u8.float().div(255)
            if kind == 'u8norm':
                xf = xf.sub(mean.view(1, 3, 1, 1)).div(std.view(1, 3, 1, 1))
            arg, is_u8, norm = u8.to(DEV), 1, (nrm if kind == 'u8norm' else None)
        conv = F.conv2d(xf.half().double(), W0.half().double(), stride=2, padding=3)
        ref = F.max_pool2d(torch.relu(conv * sc0.double().view(1, -1, 1, 1) + sh0.double().view(1, -1, 1, 1)), 3, 2, 1)
        out = torch.full((n * S * S, 96), 7.0, device=DEV, dtype=H16)
        W0d, sc0d, sh0d = W0.to(DEV), sc0.to(DEV), sh0.to(DEV)               # (kept alive: the call takes raw pointers)
        L.call('gnx_conv_stem_bnrelu_maxpool_f16mul', arg.data_ptr(), is_u8, L.ptr(W0d), out.data_ptr(), 96, n, 3, P, P, O,
               7, 7, 2, 3, L.ptr(sc0d), L.ptr(sh0d), L.ptr(norm), L.stream())
        close(out[:, :O].float().reshape(n, S, S, O).permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3, what=kind)
        assert float(out[:, O:].float().min()) == 7.0


@pytest.mark.parametrize("n,S", [(8, 4), (6, 8), (4, 16), (2, 32), (64 * 4, 32)])
def test_conv2_data_gradient_fused_with_norm2_adjoint(L, n, S):
    """gnx_conv3x3_dgrad_bnrelu_bwd == gnx_conv3x3_bnrelu (data-gradient shape) followed by gnx_bn_relu_bwd(relu = 2) - dX to
    rounding of the reciprocal (the fused kernel multiplies by 1 / scale), dgamma / dbeta to summation order."""
    g = torch.Generator().manual_seed(S * 7 + n)
    M, N, K, ld = n * S * S, 128, 32, 160
    dYfull = torch.randn(M, ld, generator=g).to(DEV)
    c0 = 96
    W2 = (torch.randn(K, N, 3, 3, generator=g) * 0.05).to(DEV)               # conv2 weight [out = 32][in = 128][3][3]
    wb = torch.empty(9, N, K, device=DEV)
    L.call('gnx_repack_conv3x3_bwd', L.ptr(W2), L.ptr(wb), K, N, L.stream())
    sc = (torch.rand(N, generator=g) + 0.5) * torch.where(torch.rand(N, generator=g) < 0.2, -1.0, 1.0)   # some negative gammas
    sh, mu, inv = torch.randn(N, generator=g) * 0.3, torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    x = torch.randn(M, N, generator=g)
    a_act = torch.relu(x * sc + sh).to(DEV)                                   # the activated bottleneck the forward stored
    scd, shd, mud, invd = sc.to(DEV), sh.to(DEV), mu.to(DEV), inv.to(DEV)
    dyp = dYfull.data_ptr() + 4 * c0
    # reference sequence
    tA = torch.empty(M, N, device=DEV)
    L.call('gnx_conv3x3_bnrelu', dyp, ld, L.ptr(wb), L.ptr(tA), N, M, N, K, S, None, None, L.stream())
    tB = torch.empty(M, N, device=DEV)
    dg, db = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    ws = torch.empty(L.query('gnx_bn_workspace', M, N), device=DEV)
    L.call('gnx_bn_relu_bwd', L.ptr(tA), N, L.ptr(a_act), N, L.ptr(tB), N, M, N, L.ptr(scd), L.ptr(shd), L.ptr(mud), L.ptr(invd),
           L.ptr(dg), L.ptr(db), 2, 0, 0, 0, L.ptr(ws), L.stream())
    # fused
    tB2 = torch.full((M, N), 7.0, device=DEV)
    dg2, db2 = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    ws2 = torch.empty(L.query('gnx_conv3x3_dgrad_bn_workspace', M, N), device=DEV)
    rc = L.query('gnx_conv3x3_dgrad_bnrelu_bwd', dyp, ld, L.ptr(wb), L.ptr(a_act), N, L.ptr(tB2), N, M, N, K, S, L.ptr(scd),
                 L.ptr(shd), L.ptr(mud), L.ptr(invd), L.ptr(dg2), L.ptr(db2), 0, L.ptr(ws2), L.stream())
    assert rc == 0
    assert torch.equal(tB2, tB)
    close(db2, db, rtol=2e-5, atol=1e-5, what='dbeta')
    close(dg2, dg, rtol=2e-4, atol=1e-4, what='dgamma')


@pytest.mark.parametrize("n,S,C", [(3, 8, 64), (2, 32, 256), (5, 4, 1024), (2, 6, 40)])
def test_transition_pooled_activation_and_pooled_adjoint(L, n, S, C):
    """Transitions run pool-first.  gnx_bnrelu_avgpool2 (the pooled, activated input) vs torch; gnx_bn_relu_bwd_pooled (norm ->
    relu adjoint straight from the pooled gradient) == gnx_avgpool2_bwd + gnx_bn_relu_bwd, bit for bit."""
    g = torch.Generator().manual_seed(S + C)
    ld = C + 8
    x = torch.randn(n * S * S, ld, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.4
    mu, inv = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    xd, scd, shd, mud, invd = x.to(DEV), sc.to(DEV), sh.to(DEV), mu.to(DEV), inv.to(DEV)
    So = S // 2
    act = torch.relu(x[:, :C] * sc + sh).reshape(n, S, S, C).permute(0, 3, 1, 2)
    want = F.avg_pool2d(act, 2, 2).permute(0, 2, 3, 1).reshape(-1, C)
    P = torch.full((n * So * So, C + 4), 7.0, device=DEV)
    L.call('gnx_bnrelu_avgpool2', L.ptr(xd), ld, L.ptr(P), C + 4, n, C, S, L.ptr(scd), L.ptr(shd), L.stream())
    close(P[:, :C], want, rtol=1e-6, atol=1e-6, what='pooled activation')
    assert float(P[:, C:].min()) == 7.0
    dYp = torch.randn(n * So * So, C, generator=g).to(DEV)
    # reference: unpool, then the adjoint
    dAct = torch.empty(n * S * S, C, device=DEV)
    L.call('gnx_avgpool2_bwd', L.ptr(dYp), C, L.ptr(dAct), C, n, C, S, L.stream())
    dx1 = torch.empty(n * S * S, ld, device=DEV)
    dg1, db1 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws = torch.empty(L.query('gnx_bn_workspace', n * S * S, C), device=DEV)
    L.call('gnx_bn_relu_bwd', L.ptr(dAct), C, L.ptr(xd), ld, L.ptr(dx1), ld, n * S * S, C, L.ptr(scd), L.ptr(shd), L.ptr(mud),
           L.ptr(invd), L.ptr(dg1), L.ptr(db1), 1, 0, 0, 0, L.ptr(ws), L.stream())
    dx2 = torch.empty(n * S * S, ld, device=DEV)
    dg2, db2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws2 = torch.empty(L.query('gnx_bn_workspace', n * S * S, C), device=DEV)
    L.call('gnx_bn_relu_bwd_pooled', L.ptr(dYp), C, L.ptr(xd), ld, L.ptr(dx2), ld, n, S, C, L.ptr(scd), L.ptr(shd), L.ptr(mud),
           L.ptr(invd), L.ptr(dg2), L.ptr(db2), 0, L.ptr(ws2), L.stream())
    assert torch.equal(dx2[:, :C], dx1[:, :C])
    # the column sums: same terms, but the two-call reference may take the single-launch small-M form (another fixed order)
    close(dg2, dg1, rtol=1e-5, atol=1e-5 * dg1.abs().max().item(), what='dgamma')
    close(db2, db1, rtol=1e-5, atol=1e-5 * db1.abs().max().item(), what='dbeta')


@pytest.mark.parametrize("M,C", [(4992, 32), (8192, 512), (5000, 1024), (2049, 100)])
def test_bn_multi_workgroup_forms_repeat_bit_for_bit(GF, M, C):
    """Round 5: the BatchNorm forms with four workgroups per channel block exchange partial sums across XCDs (whose L2s are not
    coherent) behind a barrier - a protocol error shows as an occasional STALE read, not as a wrong formula, and a stale read
    is invisible when the stale bytes happen to be the right ones (a fresh process; the same call repeated).  So: two different
    input sets alternate on the same layer (persistent self-resetting sync words), the allocator's free blocks are POISONED
    with NaN before every call (a workspace handed out again must not matter), a 256-MB fill sweeps every L2 in between; every
    call must reproduce its input set's first result bit for bit, and those results must be torch's."""
    g = torch.Generator().manual_seed(M + C)
    xs = [(torch.randn(M, C, generator=g) * 2 + 3).to(DEV), (torch.randn(M, C, generator=g) * 0.5 - 1).to(DEV)]
    dys = [torch.randn(M, C, generator=g).to(DEV), (torch.randn(M, C, generator=g) * 3).to(DEV)]
    bn = nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
    bn.train()
    xs = [_away_from_relu_kink(x.cpu(), bn.weight.detach().cpu(), bn.bias.detach().cpu(), bn.eps, True).to(DEV) for x in xs]
    junk = torch.empty(64 * 1024 * 1024, device=DEV)

    def poison():
        blocks = [torch.full((n,), float('nan'), device=DEV) for n in (256, 1024, 4096, 16384, 65536, 262144, 1048576, 4194304)
                  for _ in range(4)]
        del blocks

    first = [None, None]
    for it in range(60):
        k = it & 1
        poison()
        xd = xs[k].clone().requires_grad_(True)
        yd = GF.batch_norm_relu(xd, bn, True)
        poison()
        yd.backward(dys[k])
        cur = (yd.detach().clone(), xd.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
        bn.zero_grad()
        if first[k] is None:
            first[k] = cur
        else:
            for a, b, what in zip(cur, first[k], ('y', 'dx', 'dgamma', 'dbeta')):
                assert torch.equal(a, b), "%s differs in call %d" % (what, it)
        if it % 3 == 0:
            junk.fill_(float(it))                                   # 256 MB through every L2 between calls
    assert getattr(bn, '_gnx_sync', None) is not None and int(bn._gnx_sync.abs().sum().item()) == 0     # left zero
    for k in range(2):
        ref = nn.BatchNorm1d(C).double()                                 # (float64: see test_bn_relu_fwd_bwd)
        with torch.no_grad():
            ref.weight.copy_(bn.weight.cpu())
            ref.bias.copy_(bn.bias.cpu())
        ref.train()
        xr = xs[k].cpu().double().requires_grad_(True)
        yr = torch.relu(ref(xr))
        yr.backward(dys[k].cpu().double())
        close(first[k][0], yr, what='y')
        close(first[k][1], xr.grad, rtol=3e-4, what='dx')
        close(first[k][2], ref.weight.grad, rtol=3e-4, atol=1e-4, what='dgamma')
