"""CPU, world_size 2 over gloo: the data-parallel layer (gridnext_amd/distributed.py) that the training
loops use.  Checks, against a single-process run over the same arrays:
  * ShardedSampler covers the dataset exactly once across ranks, same count per rank;
  * allreduce_gradients averages .grad over ranks (and tolerates a parameter without a gradient);
  * a 2-rank train_gridwise run (1 array per rank per step) gives both ranks the same weights and the
    same reported epoch losses, equal to a single-process run that accumulates the two arrays' gradients
    (the semantics documented in distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem():
    from oracle import gridnet as ogn
    from oracle.mlp import count_mlp
    torch.manual_seed(5)
    G, H, W, C = 12, 6, 5, 4
    f = count_mlp(G, C)
    m = ogn.GridNetHexOddr(f, (G,), (H, W), C, use_bn=True)
    for p in m.patch_classifier.parameters():
        p.requires_grad = False
    gen = torch.Generator().manual_seed(6)
    x = torch.randint(0, 10, (4, G, H, W), generator=gen).float()
    y = torch.randint(0, C + 1, (4, H, W), generator=gen)
    return m, x, y


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import contextlib
    import io
    from gridnext_amd import distributed as gdist
    from gridnext_amd.training import train_gridwise
    r, w, dev = gdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    # gradient averaging primitive
    lin = nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(1.0)
        lin.bias.fill_(0.0)
    lin.weight.grad = torch.full((2, 3), float(rank + 1))
    gdist.allreduce_gradients([lin.weight, lin.bias])
    assert torch.allclose(lin.weight.grad, torch.full((2, 3), 1.5))
    assert lin.bias.grad is not None and float(lin.bias.grad.abs().max()) == 0.0
    # ... through ONE persistent flat buffer: .grad is a view of it (no cat, no copy-back), reused by the next step
    flat_ptr = lin.weight.grad.data_ptr()
    assert lin.bias.grad.data_ptr() == flat_ptr + 4 * lin.weight.numel()
    lin.weight.grad = None
    lin.bias.grad = torch.full((2,), float(10 * (rank + 1)))
    gdist.allreduce_gradients([lin.weight, lin.bias])
    assert lin.weight.grad.data_ptr() == flat_ptr and float(lin.weight.grad.abs().max()) == 0.0
    assert torch.allclose(lin.bias.grad, torch.full((2,), 15.0))
    # bucketed all-reduce from inside a backward (BackwardReducer): an autograd node that finishes its parameter gradients
    # in two stages hands over averaged gradients; the step's flat all-reduce then skips those parameters
    a, b, c = (nn.Parameter(torch.ones(3)), nn.Parameter(torch.ones(2)), nn.Parameter(torch.ones(4)))

    class Staged(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, a, b):
            return x.sum() + a.sum() + b.sum()

        @staticmethod
        def backward(ctx, g):
            red = gdist.BackwardReducer()
            assert gdist.BackwardReducer.wanted()
            ga, gb = torch.full((3,), float(rank + 1)), torch.full((2,), float(4 * (rank + 1)))
            red.bucket([gb], [b])                                  # "block 4" first ...
            red.bucket([ga], [a])                                  # ... "block 1" last
            assert red.finish() == 2
            return None, ga, gb

    for _ in range(2):                                             # two micro-batches accumulate (accum_iters semantics)
        (Staged.apply(torch.ones(2), a, b) + (c * float(rank + 1)).sum()).backward()
    assert torch.allclose(a.grad, torch.full((3,), 3.0)) and torch.allclose(b.grad, torch.full((2,), 12.0))
    assert torch.allclose(c.grad, torch.full((4,), 2.0 * (rank + 1)))          # not reduced yet
    gdist.allreduce_gradients([a, b, c])
    assert torch.allclose(a.grad, torch.full((3,), 3.0)) and torch.allclose(b.grad, torch.full((2,), 12.0))   # skipped
    assert torch.allclose(c.grad, torch.full((4,), 3.0))
    gdist.allreduce_gradients([a])                                 # the skip list does not outlive the step
    assert torch.allclose(a.grad, torch.full((3,), 3.0))
    # sampler
    ds = TensorDataset(torch.arange(7))
    mine = list(gdist.ShardedSampler(ds))
    assert len(mine) == 4 and mine == list(range(7))[rank::2] + ([0] if rank == 1 else [])
    # broadcast_module: rank 0's values everywhere, version counters bumped, derived-weight caches dropped (a rank that ran
    # a forward before the broadcast must not keep folded / repacked weights of its own initialisation)
    import gridnext_amd as ga
    torch.manual_seed(100 + rank)
    dn = ga.DenseNet(growth_rate=4, block_config=(2,), num_init_features=8, bn_size=2, num_classes=3, small_inputs=True)
    dn._cache['fold'] = ('stale',)
    w = dn.features.conv0.weight
    v0, e0 = w._version, dn._cache_epoch
    gdist.broadcast_module(dn)
    assert dn._cache == {} and dn._cache_epoch > e0 and w._version > v0
    ref = [torch.zeros_like(w) for _ in range(world)]
    torch.distributed.all_gather(ref, w.detach())
    assert torch.equal(ref[0], ref[1])
    assert dn.features.denseblock1.denselayer1.norm1.num_batches_tracked.dtype == torch.int64
    # epoch statistics with a padded tail: 3 items over 2 ranks -> 4 items processed, the divisor follows
    from gridnext_amd.training import _PhaseMeter
    meter = _PhaseMeter(dev)
    for _ in gdist.ShardedSampler(TensorDataset(torch.arange(3))):
        meter.add(2.0, 1, 1, 1)
    loss_sum, _, _, seen = meter.totals()
    assert seen == 4 and loss_sum / meter.n_items(3, seen) == 2.0
    # the loop
    m, x, y = _make_problem()
    gdist.broadcast_module(m)
    data = TensorDataset(x, y)
    dl = {'train': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data)),
          'val': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data))}
    opt = torch.optim.SGD(m.corrector.parameters(), lr=0.05)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        m, vh, th = train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    torch.save({'state': m.state_dict(), 'vh': vh, 'th': th, 'printed': buf.getvalue()},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_training_matches_gradient_accumulation(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / 'rank0.pt'))
    r1 = torch.load(str(tmp_path / 'rank1.pt'))
    for k in r0['state']:
        assert torch.equal(r0['state'][k], r1['state'][k]) or 'running' in k or 'num_batches' in k, k
    assert r0['th'] == r1['th'] and r0['vh'] == r1['vh']
    assert 'Loss:' in r0['printed'] and r1['printed'].strip() == ''            # rank 0 reports
    # single-process equivalent: step every 2 arrays with the MEAN of the two per-array gradients
    torch.set_num_threads(1)
    from oracle import masked_ce as oce
    m, x, y = _make_problem()
    opt = torch.optim.SGD(m.corrector.parameters(), lr=0.05)
    hist = []
    for epoch in range(2):
        m.train()
        m.patch_classifier.eval()
        total = 0.0
        for pair in ((0, 1), (2, 3)):
            opt.zero_grad()
            for i in pair:
                loss, _, _ = oce.masked_ce(m(x[i:i + 1]), y[i:i + 1], 1)
                (loss / 2).backward()
                total += loss.item()
            opt.step()
        hist.append(total / 4)
    np.testing.assert_allclose(r0['th'], hist, rtol=1e-5)
    for k, v in m.corrector.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            continue                      # BN statistics stay per-rank (documented)
        assert torch.allclose(r0['state']['corrector.' + k], v, rtol=1e-4, atol=1e-6), k


# ---------------------------------------------------------------------------------------------- exact B = world emulation
def _make_mm_problem():
    from oracle import densenet as odn, gridnet as ogn
    from oracle.mlp import count_mlp
    torch.manual_seed(15)
    G, H, W, C, P = 10, 5, 4, 3, 16
    f_img = odn.DenseNet(growth_rate=4, block_config=(2,), num_init_features=8, bn_size=2, num_classes=C, small_inputs=True)
    m = ogn.GridNetHexMM(f_img, count_mlp(G, C), (3, P, P), (G,), (H, W), C)
    for p in m.patch_classifier.parameters():             # the tutorial's recipe: freezes the IMAGE f only (the MM quirk)
        p.requires_grad = False
    gen = torch.Generator().manual_seed(16)
    xi = torch.rand((4, H, W, 3, P, P), generator=gen)
    xc = torch.randint(0, 10, (4, G, H, W), generator=gen).float()
    y = torch.randint(0, C + 1, (4, H, W), generator=gen)
    y[1, :3] = 0                                          # unequal foreground counts: the two weightings differ
    return m, xi, xc, y


class _MMData(torch.utils.data.Dataset):
    def __init__(self, xi, xc, y):
        self.xi, self.xc, self.y = xi, xc, y

    def __len__(self):
        return self.y.shape[0]

    def __getitem__(self, i):
        return (self.xi[i], self.xc[i]), self.y[i]


def _sync_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import contextlib
    import io
    from gridnext_amd import distributed as gdist
    from gridnext_amd.training import train_gridwise
    gdist.init_from_env(backend='gloo')
    m, xi, xc, y = _make_mm_problem()
    gdist.broadcast_module(m)
    gdist.convert_sync_batchnorm(m.corrector)
    gdist.convert_sync_batchnorm(m.count_classifier)
    gdist.set_sync_batchnorm(True)
    data = _MMData(xi, xc, y)
    dl = {'train': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data)),
          'val': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data))}
    params = list(m.corrector.parameters()) + list(m.count_classifier.parameters())
    opt = torch.optim.SGD(params, lr=0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        m, vh, th = train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    torch.save({'state': m.state_dict(), 'vh': vh, 'th': th}, os.path.join(out_dir, 'sync_rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sync_batchnorm_equals_one_batch_of_two_arrays(tmp_path):
    """SURVEY 8e's exact B = world emulation: with `convert_sync_batchnorm` + `set_sync_batchnorm(True)` two ranks holding
    one array each take the same steps as ONE process on batches of two arrays - train-mode BatchNorm statistics over both
    arrays (g's BatchNorm2d(32) pair and, by the GridNetHexMM quirk, the count MLP's BatchNorm1d), the loss a mean over the
    foreground spots of both (unequal counts here) - weights, running statistics and reported histories included."""
    import contextlib
    import io
    port = _free_port()
    mp.spawn(_sync_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / 'sync_rank0.pt'))
    r1 = torch.load(str(tmp_path / 'sync_rank1.pt'))
    for k in r0['state']:
        assert torch.equal(r0['state'][k], r1['state'][k]), k            # running statistics too: they are global now
    torch.set_num_threads(1)
    from gridnext_amd.training import train_gridwise
    m, xi, xc, y = _make_mm_problem()
    data = _MMData(xi, xc, y)
    dl = {'train': DataLoader(data, batch_size=2), 'val': DataLoader(data, batch_size=2)}
    params = list(m.corrector.parameters()) + list(m.count_classifier.parameters())
    opt = torch.optim.SGD(params, lr=0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        m, vh, th = train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    np.testing.assert_allclose(r0['th'], th, rtol=2e-5)
    np.testing.assert_allclose(r0['vh'], vh, rtol=2e-5)
    for k, v in m.state_dict().items():
        assert torch.allclose(r0['state'][k].float(), v.float(), rtol=2e-4, atol=2e-6), k
