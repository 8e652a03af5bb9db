"""Sustained rate of the fused fp16 stem (gnx_conv_stem_bnrelu_maxpool_f16mul) on one array of `spots` patches of P px."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
spots = int(sys.argv[1]) if len(sys.argv) > 1 else 4992
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for P in (256, 128):
    HP = P // 4
    w = (torch.randn(64, 3, 7, 7, device=DEV) * 0.1)
    sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1
    out = torch.empty(spots * HP * HP, 256, device=DEV, dtype=torch.float16)
    for u8 in (0, 1):
        x = (torch.rand(spots, 3, P, P, device=DEV) * 255).to(torch.uint8) if u8 else torch.rand(spots, 3, P, P, device=DEV)
        def run():
            L.call('gnx_conv_stem_bnrelu_maxpool_f16mul', x.data_ptr(), u8, L.ptr(w), out.data_ptr(), 256, spots, 3, P, P, 64, 7, 7, 2, 3,
                   L.ptr(sc), L.ptr(sh), None, L.stream())
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        fl = 2.0 * spots * (P // 2) ** 2 * 64 * 147
        print("P %3d %s: %7.3f ms  %6.1f TFLOP/s" % (P, 'u8 ' if u8 else 'f32', dt * 1e3, fl / dt / 1e12), flush=True)
        del x
