#!/usr/bin/env python3
"""Diagnostic for the k-split fused dense layer (csrc/dense_layer_f16_ks.hip): one small case per map size, the tape (activated
bottleneck) and the output against the double-precision reference, with the error broken down by image row / x / channel so that a
wrong tile mapping shows as a pattern.   python tools/diag/dl_ks_dbg.py [S n K]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gridnext_amd import _lib as L   # noqa: E402

DEV = 'cuda:0'
H = torch.float16


def run(S, n, K):
    ct = K + 64
    g = torch.Generator().manual_seed(S * 1000 + K + n)
    x = torch.randn(n, S, S, ct, generator=g).half()
    x[..., K:] = 7.0
    W1 = torch.randn(128, K, generator=g) * (1.0 / K ** 0.5)
    W2 = torch.randn(32, 128, 3, 3, generator=g) * 0.05
    sc1, sh1 = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    sc2, sh2 = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.5
    st = L.stream()
    L.call('gnx_dense_layer_f16_set_form', 1)
    X = x.reshape(-1, ct // 32, 32).permute(1, 0, 2).contiguous().to(DEV)
    rows = X.shape[1]
    w1p = torch.empty(128 * K, device=DEV, dtype=H)
    w2p = torch.empty(9 * 8 * 512, device=DEV, dtype=H)
    W1d, W2d = W1.to(DEV), W2.to(DEV)
    L.call('gnx_dense_layer_f16_pack', L.ptr(W1d), L.ptr(W2d), L.ptr(w1p, H), L.ptr(w2p, H), K, st)
    d = [v.to(DEV) for v in (sc1, sh1, sc2, sh2)]
    At = torch.full((4, rows, 32), 3.0, device=DEV, dtype=H)
    L.call('gnx_dense_layer_f16_tape', L.ptr(X, H), rows, n, S, K, L.ptr(w1p, H), L.ptr(w2p, H), L.ptr(d[0]), L.ptr(d[1]),
           L.ptr(d[2]), L.ptr(d[3]), L.ptr(At, H), rows, st)
    torch.cuda.synchronize()
    got = X.cpu().permute(1, 0, 2).reshape(n, S, S, ct)
    xa = torch.relu(torch.addcmul(sh1, x[..., :K].float(), sc1)).half()
    y = torch.einsum('nyxk,ok->nyxo', xa.double(), W1.half().double())
    b = torch.relu(y.float() * sc2 + sh2).half()
    a_got = At.permute(1, 0, 2).reshape(n, S, S, 128).cpu()
    ea = (a_got.double() - b.double()).abs()
    print("S=%d n=%d K=%d: tape max err %.3e (range %.2f)" % (S, n, K, ea.max().item(), b.abs().max().item()))
    if ea.max().item() > 2e-3 * b.abs().max().item():
        print("   by image:", ea.amax((1, 2, 3)).tolist()[:8])
        print("   by row  :", [round(v, 3) for v in ea.amax((0, 2, 3)).tolist()])
        print("   by x    :", [round(v, 3) for v in ea.amax((0, 1, 3)).tolist()])
        print("   by chan :", [round(v, 3) for v in ea.amax((0, 1, 2)).tolist()])
    # conv2 from the KERNEL's own bottleneck (isolates conv2 + exchange from conv1)
    ref2 = F.conv2d(a_got.double().permute(0, 3, 1, 2), W2.half().double(), padding=1).permute(0, 2, 3, 1)
    out = got[..., K:K + 32].double()
    eo = (out - ref2).abs()
    print("   output vs conv2 of the kernel's own tape: max err %.3e (range %.2f); input columns intact: %s; tail intact: %s" %
          (eo.max().item(), ref2.abs().max().item(), torch.equal(got[..., :K], x[..., :K]),
           float(got[..., K + 32:].float().min()) == 7.0 and float(got[..., K + 32:].float().max()) == 7.0))
    if eo.max().item() > 3e-3 * ref2.abs().max().item():
        print("   by image:", [round(v, 3) for v in eo.amax((1, 2, 3)).tolist()[:8]])
        print("   by row  :", [round(v, 3) for v in eo.amax((0, 2, 3)).tolist()])
        print("   by x    :", [round(v, 3) for v in eo.amax((0, 1, 3)).tolist()])
        print("   by chan :", [round(v, 3) for v in eo.amax((0, 1, 2)).tolist()])
        # which single-tap model explains it?  compare with conv2 restricted to one dy row / one dx column
        for name, mask in (("dy=-1 only", (0, None)), ("dy=0 only", (1, None)), ("dy=+1 only", (2, None)),
                           ("dx=-1 only", (None, 0)), ("dx=0 only", (None, 1)), ("dx=+1 only", (None, 2))):
            Wm = torch.zeros_like(W2)
            if mask[0] is not None:
                Wm[:, :, mask[0], :] = W2[:, :, mask[0], :]
            else:
                Wm[:, :, :, mask[1]] = W2[:, :, :, mask[1]]
            part = F.conv2d(a_got.double().permute(0, 3, 1, 2), Wm.half().double(), padding=1).permute(0, 2, 3, 1)
            print("      missing-%s residual: %.3e" % (name, ((ref2 - part) - out).abs().max().item()))


if __name__ == '__main__':
    if len(sys.argv) == 4:
        run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        for S, n, K in ((32, 1, 64), (32, 2, 128), (64, 1, 64), (64, 2, 96)):
            run(S, n, K)
