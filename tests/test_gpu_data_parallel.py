"""The data-parallel layer THROUGH THE HIP PATH on the one GPU a test box has:
  * backend 'nccl' (= RCCL), world 1, collectives forced on (GNX_DP_FORCE=1): broadcast_module, the flat gradient
    all-reduce and the epoch-statistics all-reduce all run through RCCL on device tensors; results must equal a plain
    single-process run;
  * backend 'gloo', 2 ranks sharing card 0 (GNX_DEVICE_INDEX=0): a real 2-rank train_gridwise over GridNetHexMM
    (tiny DenseNet + count MLP + hex g, fused masked CE), equal to a single-process HIP run that accumulates the two
    arrays' mean gradients, and - each rank starts from different weights with warm derived-weight caches - the
    post-broadcast forward must be rank 0's, not a stale one.
Workers are child processes (tests/dp_gpu_worker.py); nothing here execs over a process that holds the GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
WORKER = os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, backend, out_dir, extra_env):
    port = str(_free_port())
    os.makedirs(str(out_dir), exist_ok=True)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
        procs.append(subprocess.Popen([sys.executable, WORKER, str(out_dir), backend], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=420)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(os.path.join(str(out_dir), 'rank%d.pt' % r)) for r in range(world)]


def _single_process_reference(pairs, fopt=''):
    """The HIP path in this process: rank 0's initial weights, one optimizer step per group of arrays with the MEAN of the
    per-array gradients (what averaging over ranks computes).  `fopt`: f is trained too (a second SGD, as the worker's)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_gpu_worker as wk
    from gridnext_amd import functional as GF
    m, xi, xc, y = wk.make_problem(1000, fopt)
    m.to(DEV)
    f_opt = None
    if fopt:
        f_opt = torch.optim.SGD(list(m.image_classifier.parameters()) + list(m.count_classifier.parameters()), lr=0.01)
    else:
        for p in m.patch_classifier.parameters():
            p.requires_grad = False
    opt = torch.optim.SGD(m.corrector.parameters(), lr=0.05)
    hist = []
    for epoch in range(2):
        m.train()
        m.patch_classifier.eval()
        total = 0.0
        for group in pairs:
            opt.zero_grad()
            if f_opt is not None:
                f_opt.zero_grad()
            for i in group:
                logits = m.forward_nhwc([xi[i:i + 1].to(DEV), xc[i:i + 1].to(DEV)])
                loss, _, _ = GF.masked_cross_entropy(logits.reshape(-1, wk.C), y[i:i + 1].to(DEV), 1)
                (loss / len(group)).backward()
                total += loss.item()
            opt.step()
            if f_opt is not None:
                f_opt.step()
        hist.append(total / 4)
    return m, hist


@pytest.mark.timeout(600)
def test_rccl_world1_collectives_on_the_hip_path(tmp_path):
    (r0,) = _run_ranks(1, 'nccl', tmp_path, {'GNX_DP_FORCE': '1'})
    assert r0['backend'] == 'nccl' and r0['world'] == 1
    assert torch.equal(r0['before'], r0['after'])                  # broadcast from itself: same weights, same forward
    m, hist = _single_process_reference([(0,), (1,), (2,), (3,)])
    np.testing.assert_allclose(r0['th'], hist, rtol=1e-5)
    for k, v in m.corrector.state_dict().items():
        assert torch.allclose(r0['state']['corrector.' + k], v.cpu(), rtol=1e-4, atol=1e-6), k


@pytest.mark.timeout(600)
def test_two_gloo_ranks_on_one_card_through_the_hip_path(tmp_path):
    r0, r1 = _run_ranks(2, 'gloo', tmp_path, {'GNX_DEVICE_INDEX': '0'})
    assert r0['th'] == r1['th'] and r0['vh'] == r1['vh']
    assert 'Loss:' in r0['printed'] and r1['printed'].strip() == ''
    for k in r0['state']:
        if 'running' in k or 'num_batches' in k:
            continue                                               # BN statistics stay per rank (documented)
        assert torch.equal(r0['state'][k], r1['state'][k]), k
    # rank 1 began with other weights and warm caches: after broadcast_module its forward is rank 0's
    assert not torch.equal(r1['before'], r1['after'])
    assert torch.equal(r0['after'], r1['after'])
    assert torch.equal(r0['before'], r0['after'])
    m, hist = _single_process_reference([(0, 1), (2, 3)])
    np.testing.assert_allclose(r0['th'], hist, rtol=2e-5)
    for k, v in m.corrector.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            continue
        assert torch.allclose(r0['state']['corrector.' + k], v.cpu(), rtol=1e-4, atol=1e-6), k


# ---- f TRAINED under the process group: the DenseNet's HIP backward hands its gradients to
#      distributed.BackwardReducer block by block (densenet_train.py / densenet_train_f16.py) - asynchronous buckets issued
#      from inside the autograd node, waited for, divided and written back before the node returns
@pytest.mark.timeout(900)
@pytest.mark.parametrize("fopt", ["f32", "f16"])
def test_rccl_world1_f_trained_equals_the_plain_run_bit_for_bit(tmp_path, fopt):
    """RCCL, world 1, collectives forced on, f stepped by f_opt: every gradient bucket goes through an RCCL all-reduce from
    inside the backward (sum over one rank, / 1) - parameters after two epochs equal the run without a process group bit for
    bit, on the fp32 gradient path and on the fp16-MFMA one."""
    (plain,) = _run_ranks(1, 'none', tmp_path / 'plain', {'GNX_TEST_FOPT': fopt})
    (dp,) = _run_ranks(1, 'nccl', tmp_path / 'dp', {'GNX_TEST_FOPT': fopt, 'GNX_DP_FORCE': '1'})
    assert plain['backend'] == 'none' and not plain['reducer_wanted']
    assert dp['backend'] == 'nccl' and dp['reducer_wanted']           # the reducer is what ran inside the backward
    if fopt == 'f16':
        assert plain['f16_path_ran'] and dp['f16_path_ran'] and plain['overflow'] == 0 and dp['overflow'] == 0
    assert dp['th'] == plain['th'] and dp['vh'] == plain['vh']
    moved = 0
    for k in plain['state']:
        assert torch.equal(dp['state'][k], plain['state'][k]), k
    # ... and f did train: its weights are not the initial ones
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_gpu_worker as wk
    m0, _, _, _ = wk.make_problem(1000, fopt)
    for k, v in m0.state_dict().items():
        if k.startswith('image_classifier.') and k.endswith('conv1.weight'):
            moved += int(not torch.equal(v, dp['state'][k]))
    assert moved > 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fopt", ["f32", "f16"])
def test_two_gloo_ranks_f_trained_equal_single_process_accumulation(tmp_path, fopt):
    """Two ranks (gloo) sharing card 0, f trained: each rank's DenseNet backward averages its buckets with the other's inside
    the node; the result equals ONE process accumulating the two arrays' mean gradients, to 1e-6 (fp32 summation order; on the
    fp16 path the two runs' loss scales differ by the exact factor 2 of the accumulation's loss / 2)."""
    r0, r1 = _run_ranks(2, 'gloo', tmp_path, {'GNX_DEVICE_INDEX': '0', 'GNX_TEST_FOPT': fopt})
    assert r0['reducer_wanted'] and r1['reducer_wanted']
    assert r0['th'] == r1['th'] and r0['vh'] == r1['vh']
    for k in r0['state']:
        if 'running' in k or 'num_batches' in k:
            continue
        assert torch.equal(r0['state'][k], r1['state'][k]), k
    m, hist = _single_process_reference([(0, 1), (2, 3)], fopt)
    np.testing.assert_allclose(r0['th'], hist, rtol=2e-5)
    worst = 0.0
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k or not v.dtype.is_floating_point:
            continue
        a, b = r0['state'][k].double(), v.cpu().double()
        worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)))
    assert worst <= 1e-6, worst
