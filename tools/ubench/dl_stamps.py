#!/usr/bin/env python3
"""Where a step of gnx_dense_layer_f16 spends its cycles (diagnostic; runs on the GPU box).

Compiles csrc/dense_layer_f16.hip with -DGNX_DL_STAMP into tools/ubench/build/libdl_stamp.so (the product library has no
stamps), runs one launch per shape and prints, per step and averaged over workgroups, the shader cycles wave 0 (consumer)
(back group: conv2) and wave 4 (front group: conv1) spent in each segment.
"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, 'gridnext_amd', 'csrc', 'dense_layer_f16.hip')
SRC_KS = os.path.join(ROOT, 'gridnext_amd', 'csrc', 'dense_layer_f16_ks.hip')    # (the launcher dense_layer_f16.hip links to)
OUT = os.path.join(HERE, 'build', 'libdl_stamp.so')


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DGNX_DL_STAMP',
                    '-I', os.path.dirname(SRC), SRC, SRC_KS, '-o', OUT], check=True)
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
        X = torch.randn(ct // 32, M, 32, device=dev).to(H)                  # channel-blocked [ct / 32][rows][32]
        W1 = torch.randn(128, K, device=dev) / K ** 0.5
        W2 = torch.randn(32, 128, 3, 3, device=dev) * 0.05
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        osc, osh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
        w1p, w2p = torch.empty(128 * K, device=dev, dtype=H), torch.empty(9 * 8 * 512, device=dev, dtype=H)
        lib.gnx_dense_layer_f16_pack(W1.data_ptr(), W2.data_ptr(), w1p.data_ptr(), w2p.data_ptr(), K, st)

        def launch():
            lib.gnx_dense_layer_f16(X.data_ptr(), M, n, S, K, w1p.data_ptr(), w2p.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                    osc.data_ptr(), osh.data_ptr(), st)
        units = n if S >= 16 else M // 128
        J = S * S // 128 if S >= 16 else 1
        grid = min(units, 256)
        steps = -(-units // grid) * J                           # steps of the busiest workgroup
        stamps = torch.zeros(grid * 24, device=dev, dtype=torch.int64)
        lib.gnx_dense_layer_f16_set_stamps(stamps.data_ptr(), 0)
        launch()
        launch()
        torch.cuda.synchronize()
        v = stamps.view(grid, 24).double().mean(0) / steps
        print("S=%2d K=%4d cycles/step: consumer wave 0 [barrier wait %.0f | conv1 %.0f | epilogue %.0f | E wait %.0f | conv2 taps %.0f | stores %.0f]"
              "  feeder wave 4 [barrier wait %.0f | data wait %.0f | apply %.0f | issue %.0f | E wait %.0f | conv2 %.0f | stores %.0f | first apply %.0f]"
              % (S, K, v[0], v[1], v[2], v[3], v[4], v[5], v[8], v[15], v[9], v[10], v[11], v[12], v[13], v[14]), flush=True)
        # what the launch costs when a part is left out (wrong results, timing only)
        names = {0: 'full', 127: 'sync only', 127 + 512: 'sync, no x load issue', 127 + 1024: 'sync, no consts', 127 + 1536: 'sync, neither',
                 512 + 4: 'full, no x load issue', 1024: 'full, no consts'}
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
        del X


if __name__ == '__main__':
    main()
