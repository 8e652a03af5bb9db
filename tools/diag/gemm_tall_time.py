"""Times the composed first layer of the count MLP over one 78 x 64 grid (M = 4992 spots, K = 2000 genes K-major, N = 100) with
and without the K split of the 64 x 64 GEMM kernel (workspace given / not given), and the 2000 -> 500 layer for reference."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gridnext_amd import _lib as L   # noqa: E402

DEV = 'cuda:0'


def timeit(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for M, K, N in ((4992, 2000, 100), (4992, 2000, 500), (4992, 2000, 128), (4992, 2000, 64)):
    x = torch.randint(0, 10, (K, M), device=DEV).float()
    w = torch.randn(N, K, device=DEV) * 0.05
    b = torch.randn(N, device=DEV)
    y = torch.empty(M, N, device=DEV)
    nws = L.query('gnx_gemm_f32_workspace', M, N, K)
    ws = torch.empty(max(nws, 1), device=DEV)
    st = L.stream()
    t_ws = timeit(lambda: L.call('gnx_gemm_f32_ws', L.ptr(x), M, 1, L.ptr(w), K, 0, L.ptr(b), L.ptr(y), N, M, N, K, 0, L.ptr(ws), st))
    t_no = timeit(lambda: L.call('gnx_gemm_f32', L.ptr(x), M, 1, L.ptr(w), K, 0, L.ptr(b), L.ptr(y), N, M, N, K, 0, st))
    print("M=%d K=%d N=%d: workspace %d floats; with workspace %.1f us, without %.1f us (%.1f TFLOP/s)" %
          (M, K, N, nws, t_ws, t_no, 2.0 * M * K * N / t_ws / 1e6))
