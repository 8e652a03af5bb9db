"""gnx_hexconv_bwd_weight_batch against the single calls on a tiny grid (8 x 6) and a Visium grid, five layers."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gridnext_amd import _lib as L, functional as GF
DEV = 'cuda:0'
for (B, H, W) in ((1, 8, 6), (1, 78, 64), (2, 9, 7)):
    layers = [(5, 32), (32, 32), (32, 32), (32, 32), (32, 5)]
    g = torch.Generator().manual_seed(H)
    items, single = [], []
    for (I, O) in layers:
        x = torch.randn(B, H, W, I, generator=g).to(DEV)
        dy = torch.randn(B, H, W, O, generator=g).to(DEV)
        out = []
        for _ in range(2):
            dk0 = torch.full((O, I, 3, 1), 7.0, device=DEV)
            dk1 = torch.full((O, I, 2, 2), 7.0, device=DEV)
            db = torch.full((O,), 7.0, device=DEV)
            out.append((dk0, dk1, db))
        ws = torch.empty(L.query('gnx_hexconv_bwd_weight_workspace', B, H, W, I, O), device=DEV)
        L.call('gnx_hexconv_bwd_weight', L.ptr(x), L.ptr(dy), L.ptr(out[0][0]), L.ptr(out[0][1]), L.ptr(out[0][2]), L.ptr(ws), B, H, W, I, O, 1, 0, L.stream())
        items.append((x, dy) + out[1] + (B, H, W, I, O, 1, None))
        single.append(out[0])
    arr = (GF._HexWgradItem * len(items))()
    keep = []
    for a, (x, dy, dk0, dk1, db, B_, H_, W_, I, O, mode, _k) in zip(arr, items):
        ws = torch.empty(L.query('gnx_hexconv_bwd_weight_workspace', B_, H_, W_, I, O), device=DEV)
        keep.append(ws)
        a.x, a.dy, a.dkernel0, a.dkernel1, a.dbias, a.workspace = L.ptr(x), L.ptr(dy), L.ptr(dk0), L.ptr(dk1), L.ptr(db), L.ptr(ws)
        a.B, a.H, a.W, a.I, a.O, a.mode, a.accumulate, a.pad = B_, H_, W_, I, O, mode, 0, 0
    rc = L.query('gnx_hexconv_bwd_weight_batch', ctypes.addressof(arr), len(items), L.stream())
    torch.cuda.synchronize()
    print("grid", (B, H, W), "rc", rc, [tuple(torch.equal(a, b) for a, b in zip(s, it[2:5])) for s, it in zip(single, items)])
