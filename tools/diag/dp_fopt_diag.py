#!/usr/bin/env python3
"""Diagnostic: ONE data-parallel step (2 gloo ranks on card 0, f trained) vs the single-process mean of the two arrays'
gradients, per parameter.   python tools/diag/dp_fopt_diag.py [f32|f16]      (workers: same script with --worker)"""
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def one_step_grads(m, xi, xc, y, idx, dev, scale):
    import torch
    from gridnext_amd import functional as GF
    import dp_gpu_worker as wk
    m.train()
    m.patch_classifier.eval()
    for i in idx:
        logits = m.forward_nhwc([xi[i:i + 1].to(dev), xc[i:i + 1].to(dev)])
        loss, _, _ = GF.masked_cross_entropy(logits.reshape(-1, wk.C), y[i:i + 1].to(dev), 1)
        (loss * scale).backward()


def two_steps(m, xi, xc, y, groups, dev, scale, reduce_fn):
    """two optimizer steps (SGD as the worker's); returns per-array losses and the parameters after each step"""
    import torch
    from gridnext_amd import functional as GF
    import dp_gpu_worker as wk
    opt = torch.optim.SGD(m.corrector.parameters(), lr=0.05)
    f_opt = torch.optim.SGD(list(m.image_classifier.parameters()) + list(m.count_classifier.parameters()), lr=0.01)
    losses, snaps = [], []
    m.train()
    m.patch_classifier.eval()
    for group in groups:
        for i in group:
            logits = m.forward_nhwc([xi[i:i + 1].to(dev), xc[i:i + 1].to(dev)])
            loss, _, _ = GF.masked_cross_entropy(logits.reshape(-1, wk.C), y[i:i + 1].to(dev), 1)
            (loss * scale).backward()
            losses.append(float(loss.item()))
        reduce_fn()
        opt.step()
        f_opt.step()
        opt.zero_grad()
        f_opt.zero_grad()
        snaps.append({n: p.detach().cpu().clone() for n, p in m.named_parameters()})
    return losses, snaps


def worker(out_dir, fopt):
    import torch
    import dp_gpu_worker as wk
    from gridnext_amd import distributed as gdist
    rank, world, dev = gdist.init_from_env(backend='gloo')
    m, xi, xc, y = wk.make_problem(1000 + rank, fopt)
    m.to(dev)
    gdist.broadcast_module(m)
    params = [p for p in m.parameters() if p.requires_grad]
    if os.environ.get('GNX_DIAG_STEPS') == '2':
        losses, snaps = two_steps(m, xi, xc, y, [[rank], [2 + rank]], dev, 1.0, lambda: gdist.allreduce_gradients(params))
        torch.save({'losses': losses, 'snaps': snaps}, os.path.join(out_dir, 'diag%d.pt' % rank))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        return
    one_step_grads(m, xi, xc, y, [rank], dev, 1.0)
    raw = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    early = set(gdist._EARLY)
    gdist.allreduce_gradients(params)
    torch.cuda.synchronize()
    red = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    torch.save({'raw': raw, 'red': red, 'early': [n for n, p in m.named_parameters() if id(p) in early]},
               os.path.join(out_dir, 'diag%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main():
    fopt = sys.argv[1] if len(sys.argv) > 1 else 'f32'
    if '--worker' in sys.argv:
        return worker(sys.argv[sys.argv.index('--worker') + 1], fopt)
    import torch
    out_dir = tempfile.mkdtemp()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                   GNX_DEVICE_INDEX='0', GNX_TEST_FOPT=fopt, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), fopt, '--worker', out_dir], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    d0, d1 = (torch.load(os.path.join(out_dir, 'diag%d.pt' % r)) for r in range(2))
    os.environ['GNX_TEST_FOPT'] = fopt
    import dp_gpu_worker as wk
    m, xi, xc, y = wk.make_problem(1000, fopt)
    dev = torch.device('cuda:0')
    m.to(dev)
    if os.environ.get('GNX_DIAG_STEPS') == '2':
        losses, snaps = two_steps(m, xi, xc, y, [[0, 1], [2, 3]], dev, 0.5, lambda: None)
        print("overlap", os.environ.get('GNX_DP_OVERLAP', '1'), "losses reference", losses, "rank0", d0['losses'], "rank1", d1['losses'])
        for k in range(2):
            worst = sorted(((float((d0['snaps'][k][n].double() - snaps[k][n].double()).abs().max() /
                                   snaps[k][n].double().abs().max().clamp_min(1e-30)), n) for n in snaps[k]), reverse=True)[:6]
            print("after step %d: worst parameter differences (rel):" % (k + 1), ["%.2e %s" % w for w in worst])
        return
    one_step_grads(m, xi, xc, y, [0, 1], dev, 0.5)
    ref = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    print("early-reduced parameters on rank 0: %d of %d" % (len(d0['early']), len(ref)))
    rows = []
    for n in ref:
        a, b = d0['red'][n].double(), ref[n].double()
        mean_raw = 0.5 * (d0['raw'][n].double() + d1['raw'][n].double())
        rows.append((float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)),
                     float(a.norm() / b.norm().clamp_min(1e-30)),
                     float((mean_raw - b).abs().max() / b.abs().max().clamp_min(1e-30)),
                     float((d0['red'][n].double() - d1['red'][n].double()).abs().max()), n, n in d0['early']))
    rows.sort(reverse=True)
    print("worst 12 (rel err of reduced vs reference | norm ratio | rel err of mean(raw ranks) vs reference | rank0-rank1 | name | early)")
    for r in rows[:12]:
        print("  %.2e  %.4f  %.2e  %.2e  %s  %s" % r)
    bad = [r for r in rows if r[0] > 1e-4]
    print("parameters off by > 1e-4: %d of %d; of them early: %d" % (len(bad), len(rows), sum(1 for r in bad if r[5])))


if __name__ == '__main__':
    main()
