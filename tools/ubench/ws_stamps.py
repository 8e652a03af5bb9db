"""Reads the consumer-wave stamps of a -DGNX_WS_STAMP=1 build of conv1x1_ws_kernel (GNX_LIB=tools/ubench/build/libws_stamp.so):
how many cycles of the chunk loop the MFMA waves spend waiting at the per-chunk barrier."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
for S, K, ct in [(32, 224, 256), (16, 480, 512), (8, 992, 1024)]:
    M = 4992 * S * S
    A = torch.randn(M, ct, device=DEV)
    W = torch.randn(128, K, device=DEV) * 0.05
    out = torch.empty(M, 128, device=DEV)
    sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
    for _ in range(30):
        L.call('gnx_conv1x1_bnrelu', L.ptr(A), ct, L.ptr(W), L.ptr(out), 128, M, 128, K, L.ptr(sc), L.ptr(sh), 0, 0, L.stream())
    torch.cuda.synchronize()
    o = out.view(M // 128, 128, 128)[:, :4, :2].double().cpu()       # [tile][wave][bar,total]
    bar, tot = o[..., 0].mean().item(), o[..., 1].mean().item()
    nk = K // 32
    print("S=%2d K=%4d: chunk loop %8.0f cycles/tile (%6.0f per chunk; 64 MFMAs = 4096), at barriers %8.0f (%4.1f%%)"
          % (S, K, tot, tot / nk, bar, 100 * bar / tot))
