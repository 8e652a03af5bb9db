#!/usr/bin/env python3
"""Where a step of gnx_dense_layer_f16 spends its cycles (diagnostic; runs on the GPU box).

Compiles csrc/dense_layer_f16.hip with -DGNX_DL_STAMP into tools/ubench/build/libdl_stamp.so (the product library has no
stamps), runs one launch per shape and prints, per step and averaged over workgroups, the shader cycles wave 0 (consumer)
and wave 4 (producer) spent in each segment.
"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, 'gridnext_amd', 'csrc', 'dense_layer_f16.hip')
OUT = os.path.join(HERE, 'build', 'libdl_stamp.so')


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DGNX_DL_STAMP',
                    '-I', os.path.dirname(SRC), SRC, '-o', OUT], check=True)
    lib = ctypes.CDLL(OUT)
    P, I, Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    lib.gnx_dense_layer_f16_pack.argtypes = [P, P, P, P, I, P]
    lib.gnx_dense_layer_f16.argtypes = [P, Lg, Lg, I, I, P, P, P, P, P, P, P]
    lib.gnx_dense_layer_f16_set_stamps.argtypes = [P, I]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1664
    dev = 'cuda:0'
    H = torch.float16
    st = torch.cuda.current_stream().cuda_stream
    for S, K, ct in [(64, 64, 256), (64, 224, 256), (32, 480, 512), (16, 992, 1024), (8, 992, 1024)]:
        M = n * S * S
        X = torch.randn(M, ct, device=dev).to(H)
        W1 = torch.randn(128, K, device=dev) / K ** 0.5
        W2 = torch.randn(32, 128, 3, 3, device=dev) * 0.05
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        osc, osh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
        w1p, w2p = torch.empty(128 * K, device=dev, dtype=H), torch.empty(9 * 8 * 512, device=dev, dtype=H)
        lib.gnx_dense_layer_f16_pack(W1.data_ptr(), W2.data_ptr(), w1p.data_ptr(), w2p.data_ptr(), K, st)
        stamps = torch.zeros(256 * 24, device=dev, dtype=torch.int64)

        def launch():
            lib.gnx_dense_layer_f16(X.data_ptr(), ct, n, S, K, w1p.data_ptr(), w2p.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                    osc.data_ptr(), osh.data_ptr(), st)
        # ablations first, timed WITHOUT stamps (a NULL stamp buffer): what the launch costs when a part is left out
        names = {0: 'full', 128: 'conv2 -Wreads', 256: 'conv2 -Areads', 384: 'conv2 -reads', 1: '-norm1', 2: '-conv1', 4: '-dma', 8: '-conv2', 16: '-epilogue', 3: '-norm1-conv1', 24: '-conv2-epi',
                 27: 'dma+sync only', 31: 'sync only'}
        line = []
        for abl, nm in names.items():
            lib.gnx_dense_layer_f16_set_stamps(None, abl)
            launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                launch()
            e1.record()
            torch.cuda.synchronize()
            line.append("%s %.3f" % (nm, e0.elapsed_time(e1) / 10))
        print("S=%2d K=%4d ms: " % (S, K) + " | ".join(line), flush=True)
        lib.gnx_dense_layer_f16_set_stamps(stamps.data_ptr(), 0)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        units = n if S >= 16 else n * S * S // 128
        J = S * S // 128 if S >= 16 else 1
        t = stamps.view(256, 24).double().cpu()
        steps = torch.tensor([len(range(b, units, 256)) * J for b in range(256)], dtype=torch.double).clamp(min=1)
        per = (t / steps[:, None]).mean(0)
        nst = K // 32
        print("S=%2d K=%4d (%2d stages/step)  consumer: barrier %6.0f  stage-body %6.0f  epilogue %5.0f  E-barrier %5.0f  conv2 %6.0f | "
              "loader: barrier %6.0f  issue %5.0f  dma-wait %6.0f  E %5.0f | activator: barrier %6.0f  activate %5.0f  E %5.0f   "
              "[cycles per step]" %
              (S, K, nst, per[0], per[1], per[2], per[3], per[4], per[8], per[9], per[10], per[12], per[16], per[19], per[20]),
              flush=True)
        del X


if __name__ == '__main__':
    main()
