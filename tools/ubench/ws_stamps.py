"""Reads the wave stamps of a -DGNX_WS_STAMP=1 build of conv1x1_ws_kernel (GNX_LIB=tools/ubench/build/libws_stamp.so):
cycles the consumer (MFMA) waves and producer wave 4 of every persistent workgroup spend waiting at the per-chunk barrier."""
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
    if os.environ.get('WS_CLAMP'):
        osc, osh = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
        Wf, bounds, oshf = torch.empty(128, K, device=DEV), torch.empty(2, K, device=DEV), torch.empty(128, device=DEV)
        L.call('gnx_conv1x1_fold_clamp', L.ptr(W), L.ptr(sc), L.ptr(sh), L.ptr(osc), L.ptr(osh), L.ptr(Wf), L.ptr(bounds),
               L.ptr(oshf), 128, K, L.stream())
    for _ in range(30):
        if os.environ.get('WS_CLAMP'):
            L.call('gnx_conv1x1_clamped_act', L.ptr(A), ct, L.ptr(Wf), L.ptr(bounds), L.ptr(out), 128, M, 128, K, L.ptr(osc),
                   L.ptr(oshf), L.stream())
            continue
        L.call('gnx_conv1x1_bnrelu', L.ptr(A), ct, L.ptr(W), L.ptr(out), 128, M, 128, K, (L.ptr(sc) if os.environ.get('WS_ACT', '1') == '1' else None), (L.ptr(sh) if os.environ.get('WS_ACT', '1') == '1' else None), 0, 0, L.stream())
    torch.cuda.synchronize()
    nwg = min(M // 128, 512)
    o = out.view(M // 128, 128, 128)[:nwg, :5, :7].double().cpu()       # [workgroup = its first tile][wave 0-3, producer][..]
    cw, ct_ = o[:, :4, 0].mean().item(), o[:, :4, 1].mean().item()
    pw, pt, chunks = o[:, 4, 0].mean().item(), o[:, 4, 1].mean().item(), o[:, 4, 2].mean().item()
    print("S=%2d K=%4d: %5.0f chunks/workgroup; consumer %7.0f cycles/chunk (ideal 2 x 4096 shared by two workgroups), at "
          "barriers %4.1f%%; producer at barriers %4.1f%% (work %6.0f cycles/chunk)"
          % (S, K, chunks, ct_ / chunks, 100 * cw / ct_, 100 * pw / pt, (pt - pw) / chunks))
    seg = [o[:, 4, 3 + q].mean().item() / chunks for q in range(4)]
    print("      producer cycles/chunk: issue loads %5.0f | wait operands %5.0f | BN+ReLU + issue ds_write %5.0f | wait ds_write %5.0f"
          % tuple(seg))
