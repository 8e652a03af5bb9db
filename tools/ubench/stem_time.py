"""Times the fused stem (conv0 + norm0 + relu0 + pool0) at 128 and 256 px; GNX_LIB selects the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
for P, n in ((128, 4992), (256, 1248)):
    x = torch.rand(n, 3, P, P, device=DEV)
    W = torch.randn(64, 3, 7, 7, device=DEV) * 0.1
    sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.2
    out = torch.empty(n * (P // 4) ** 2, 256, device=DEV)
    run = lambda: L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(x), L.ptr(W), L.ptr(out), 256, n, 3, P, P, 64, 7, 7, 2, 3,
                         L.ptr(sc), L.ptr(sh), L.stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print('fused stem %d px, %d spots: %.3f ms' % (P, n, s.elapsed_time(e) / 20))
