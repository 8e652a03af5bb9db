import sys, os, ctypes
sys.path.insert(0, '/root/repo')
import torch
from gridnext_amd import _lib as L
new = L.lib()
old = ctypes.CDLL('/root/repo/tools/ubench/build/libold_bn.so')
for name, (res, args) in L.SIGNATURES.items():
    if hasattr(old, name):
        fn = getattr(old, name); fn.restype, fn.argtypes = res, args
DEV='cuda:0'
torch.manual_seed(0)
def run(lib, M, C, ld, x, dy, gamma, beta, relu, training):
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    st = torch.zeros(4, C, device=DEV)
    ws = torch.empty(lib.gnx_bn_workspace(M, C), device=DEV)
    s = L.stream()
    assert lib.gnx_bn_train_stats(L.ptr(x), ld, M, C, L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt, torch.int64), 0.1, 1e-5,
           L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]), L.ptr(ws), s) == 0
    dx = torch.empty(M, C, device=DEV); dg = torch.empty(C, device=DEV); db = torch.empty(C, device=DEV)
    assert lib.gnx_bn_relu_bwd(L.ptr(dy), C, L.ptr(x), ld, L.ptr(dx), C, M, C, L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]),
           L.ptr(dg), L.ptr(db), relu, training, 0, 0, L.ptr(ws), s) == 0
    out = torch.empty(C, device=DEV)
    assert lib.gnx_colsum(L.ptr(dy), C, M, C, L.ptr(out), 0, L.ptr(ws), s) == 0
    torch.cuda.synchronize()
    return [st.clone(), rm, rv, dx, dg, db, out]
for M, C, ld in [(32, 500, 500), (32, 100, 100), (32, 50, 50), (32, 8, 8), (32, 100, 128), (256, 100, 100), (257, 100, 100), (1000, 500, 500)]:
    xb = torch.randn(M, ld, device=DEV) * 2 + 1
    x = xb[:, :C] if ld != C else xb
    dy = torch.randn(M, C, device=DEV)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    for relu in (0, 1):
        for training in (1, 0):
            a = run(old, M, C, ld, x, dy, gamma, beta, relu, training)
            b = run(new, M, C, ld, x, dy, gamma, beta, relu, training)
            d = [float((p - q).abs().max() / (p.abs().max() + 1e-30)) for p, q in zip(a, b)]
            print(M, C, ld, 'relu', relu, 'train', training, ' '.join('%.1e' % v for v in d))
