"""Times the fp16-MFMA fused stem (config 5) at 128 and 256 px, float and uint8 patches, against the fp32 fused stem."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
H = torch.float16
def timeit(run, reps=10):
    for _ in range(2): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for P, n in ((128, 4992), (256, 1664)):
    x = torch.rand(n, 3, P, P, device=DEV)
    x8 = (x * 255).to(torch.uint8)
    W = torch.randn(64, 3, 7, 7, device=DEV) * 0.1
    sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.2
    out = torch.empty(n * (P // 4) ** 2, 256, device=DEV)
    out16 = torch.empty(n * (P // 4) ** 2, 256, device=DEV, dtype=H)
    st = L.stream()
    t32 = timeit(lambda: L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(x), L.ptr(W), L.ptr(out), 256, n, 3, P, P, 64, 7, 7, 2, 3,
                                L.ptr(sc), L.ptr(sh), st))
    t16 = timeit(lambda: L.call('gnx_conv_stem_bnrelu_maxpool_f16mul', L.ptr(x), 0, L.ptr(W), L.ptr(out16, H), 256, n, 3, P, P,
                                64, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh), None, st))
    t8 = timeit(lambda: L.call('gnx_conv_stem_bnrelu_maxpool_f16mul', x8.data_ptr(), 1, L.ptr(W), L.ptr(out16, H), 256, n, 3, P, P,
                               64, 7, 7, 2, 3, L.ptr(sc), L.ptr(sh), None, st))
    fl = 2.0 * n * (P // 2) ** 2 * 147 * 64
    print('%d px, %d spots: fp32 stem %.3f ms (%.0f TFLOP/s) | fp16-MFMA stem, float patches %.3f ms (%.0f), uint8 patches %.3f ms (%.0f)'
          % (P, n, t32, fl / t32 / 1e9, t16, fl / t16 / 1e9, t8, fl / t8 / 1e9))
