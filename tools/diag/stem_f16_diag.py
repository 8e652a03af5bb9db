"""Diagnostic: which taps of the fp16-multiply stem disagree with the fp32-multiply stem (one-hot weights)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
H = torch.float16
P, n, O = 128, 2, 64
S = P // 4
g = torch.Generator().manual_seed(0)
x = torch.rand(n, 3, P, P, generator=g).half().float().to(DEV)      # fp16-representable pixels
sc, sh = torch.ones(O, device=DEV), torch.zeros(O, device=DEV)
bad = []
for c in range(3):
    for ky in range(7):
        for kx in range(7):
            W = torch.zeros(O, 3, 7, 7, device=DEV)
            W[5, c, ky, kx] = 1.0
            a = torch.empty(n * S * S, 64, device=DEV, dtype=H)
            b = torch.empty(n * S * S, 64, device=DEV, dtype=H)
            L.call('gnx_conv_stem_bnrelu_maxpool_h16', L.ptr(x), L.ptr(W), L.ptr(a, H), 64, n, 3, P, P, O, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh), L.stream())
            L.call('gnx_conv_stem_bnrelu_maxpool_f16mul', x.data_ptr(), 0, L.ptr(W), b.data_ptr(), 64, n, 3, P, P, O, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh), None, L.stream())
            d = (a.float() - b.float()).abs().max().item()
            d5 = (a[:, 5].float() - b[:, 5].float()).abs().max().item()
            if d > 1e-3:
                bad.append((c, ky, kx, round(d, 3), round(d5, 3)))
print("bad taps:", len(bad))
print(bad[:40])
