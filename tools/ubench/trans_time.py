"""Times the three DenseNet-121 transitions (pool-first conv1x1) of one 128-px array: GNX_LIB selects the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
n = 4992
tot = 0.0
for S, K, N in [(32, 256, 128), (16, 512, 256), (8, 1024, 512)]:
    M = n * S * S
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * 0.05
    sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
    out = torch.empty(M // 4, N, device=DEV)
    run = lambda: L.call('gnx_conv1x1_bnrelu', L.ptr(A), K, L.ptr(W), L.ptr(out), N, M // 4, N, K, L.ptr(sc), L.ptr(sh), 1, S, L.stream())
    for _ in range(20): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 100
    tot += ms
    print("transition S=%2d K=%4d N=%3d  %.3f ms  %.2f TB/s in" % (S, K, N, ms, 4.0 * M * K / ms / 1e9))
    del A, out
print("sum %.3f ms" % tot)
