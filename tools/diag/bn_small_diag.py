"""Diagnostic: single-launch BatchNorm forms (M <= 8192) against the slab kernels (GNX_BN_NO_SMALL=1 in a child process)."""
import os, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from gridnext_amd import _lib as L
    DEV = 'cuda:0'
    out = {}
    for M, C, ld in ((24, 100, 100), (24, 32, 32), (32, 992, 1024), (128, 480, 512), (4992, 32, 32), (2048, 128, 128), (512, 992, 1024), (8192, 256, 512)):
        g = torch.Generator().manual_seed(M + C)
        x = (torch.randn(M, ld, generator=g) * 2 + 0.5).to(DEV)
        dy = torch.randn(M, ld, generator=g).to(DEV)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        stats = torch.empty(4, C, device=DEV)
        ws = torch.empty(L.query('gnx_bn_workspace', M, C), device=DEV)
        st = L.stream()
        L.call('gnx_bn_train_stats', L.ptr(x), ld, M, C, L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt, torch.int64),
               0.1, 1e-5, L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(ws), st)
        dx0 = torch.randn(M, ld, generator=g).to(DEV)
        for training in (1, 0):
            dx = dx0.clone()
            dg, db = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
            L.call('gnx_bn_relu_bwd', L.ptr(dy), ld, L.ptr(x), ld, L.ptr(dx), ld, M, C, L.ptr(stats[0]), L.ptr(stats[1]),
                   L.ptr(stats[2]), L.ptr(stats[3]), L.ptr(dg), L.ptr(db), 1, training, 1, 1, L.ptr(ws), st)
            out[(M, C, 'dx', training)] = dx[:, :C].cpu(); out[(M, C, 'dg', training)] = dg.cpu(); out[(M, C, 'db', training)] = db.cpu()
        out[(M, C, 'stats')] = stats.cpu(); out[(M, C, 'rm')] = rm.cpu(); out[(M, C, 'rv')] = rv.cpu(); out[(M, C, 'nbt')] = nbt.cpu()
        # fp64 reference of the statistics
        xd = x[:, :C].double()
        out[(M, C, 'ref_mean')] = xd.mean(0).cpu(); out[(M, C, 'ref_var')] = xd.var(0, unbiased=False).cpu()
    torch.save(out, sys.argv[2])
    sys.exit(0)
res = {}
for tag, env in (('small', {}), ('slab', {'GNX_BN_NO_SMALL': '1'})):
    f = '/tmp/bn_diag_%s.pt' % tag
    subprocess.run([sys.executable, os.path.abspath(__file__), 'child', f], env={**os.environ, **env}, check=True)
    res[tag] = torch.load(f)
for k in res['small']:
    a, b = res['small'][k].double(), res['slab'][k].double()
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)
    extra = ''
    if k[2] == 'stats':
        m, v = res['small'][(k[0], k[1], 'ref_mean')], res['small'][(k[0], k[1], 'ref_var')]
        extra = ' | mean err small %.2e slab %.2e; invstd err small %.2e slab %.2e' % (
            (a[2] - m).abs().max(), (b[2] - m).abs().max(),
            (a[3] - (v + 1e-5).rsqrt()).abs().max(), (b[3] - (v + 1e-5).rsqrt()).abs().max())
    print(k, 'rel diff %.3e' % err, extra)
