#!/usr/bin/env python3
"""Where a step of the k-split fused dense layer (csrc/dense_layer_f16_ks.hip) spends its cycles (diagnostic; GPU box).

Compiles the kernel file with -DGNX_KS_STAMP into tools/ubench/build/libks_stamp.so (the product library has no stamps), runs
one launch per shape and prints, per 128-pixel step and averaged over workgroups, the shader cycles waves 0 and 2 spent in each
segment.  The fragment packing comes from the product library."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from gridnext_amd import _lib as L   # noqa: E402

SRC = os.path.join(ROOT, 'gridnext_amd', 'csrc', 'dense_layer_f16_ks.hip')
OUT = os.path.join(HERE, 'build', 'libks_stamp.so')
NAMES = ['operands landed', 'norm1+write', 'W wait+MFMA issue', 'pair barrier', 'norm2/relu2', 'conv2', 'sums out+barrier',
         'finish']


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DGNX_KS_STAMP',
                    '-I', os.path.dirname(SRC), SRC, '-o', OUT], check=True)
    lib = ctypes.CDLL(OUT)
    P, I, Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    lib.gnx_ks_launch.argtypes = [P, Lg, Lg, I, I, P, P, P, P, P, P, I, P]
    lib.gnx_ks_set_stamps.argtypes = [P]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4992
    dev = 'cuda:0'
    H = torch.float16
    st = torch.cuda.current_stream().cuda_stream
    for S, K, ct in [(64, 64, 256), (64, 224, 256), (32, 128, 512), (32, 480, 512)]:
        M = n * S * S
        X = torch.randn(ct // 32, M, 32, device=dev).to(H)
        W1 = torch.randn(128, K, device=dev) / K ** 0.5
        W2 = torch.randn(32, 128, 3, 3, device=dev) * 0.05
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        osc, osh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
        w1p, w2p = torch.empty(128 * K, device=dev, dtype=H), torch.empty(9 * 8 * 512, device=dev, dtype=H)
        L.call('gnx_dense_layer_f16_pack', L.ptr(W1), L.ptr(W2), L.ptr(w1p, H), L.ptr(w2p, H), K, L.stream())
        grid = min(n, 512)
        steps = -(-n // grid) * (S * S // 128)
        stamps = torch.zeros(grid * 16, device=dev, dtype=torch.int64)
        lib.gnx_ks_set_stamps(stamps.data_ptr())

        def launch():
            lib.gnx_ks_launch(X.data_ptr(), M, n, S, K, w1p.data_ptr(), w2p.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                              osc.data_ptr(), osh.data_ptr(), 256, st)
        launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        v = stamps.view(grid, 2, 8).double().mean(0) / steps
        for w in range(2):
            print("S=%2d K=%4d wave %d cycles/step (total %.0f): " % (S, K, 2 * w, v[w].sum()) +
                  " | ".join("%s %.0f" % (NAMES[k], v[w][k]) for k in range(8)), flush=True)
        print("   launch %.3f ms = %.2f us per step and workgroup" % (e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / steps), flush=True)
        del X


if __name__ == '__main__':
    main()
