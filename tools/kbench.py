#!/usr/bin/env python3
"""Kernel micro-benchmark harness for the DenseNet conv kernels (tuning aid; runs on the GPU box).

    python tools/kbench.py [--spots 4992] [--reps 5] [--only conv3x3|conv1x1|stem|pool]
Prints per-shape time and achieved TFLOP/s (algorithmic FLOPs) using HIP events on the launch stream.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridnext_amd import _lib as L   # noqa: E402

DEV = 'cuda:0'


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--spots', type=int, default=4992)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--noact', action='store_true', help='conv3x3 without the BN+ReLU prologue')
    ap.add_argument('--const', action='store_true', help='constant operands (low bit toggling) instead of randn')
    ap.add_argument('--clamp', action='store_true', help='conv1x1 in its folded LDS-clamp form (gnx_conv1x1_clamped_act)')
    ap.add_argument('--wino', action='store_true', help='conv3x3 in its Winograd F(2,3) form (prologue-free operand)')
    ap.add_argument('--form', type=int, default=0, help='fused dense layer: 0 = LDS form, 1 = k-split form (64/32-px maps)')
    ap.add_argument('--dense', action='store_true', help='operand rows exactly K wide (lda = K) instead of the block buffer stride')
    args = ap.parse_args()
    n = args.spots
    st = L.stream()
    torch.manual_seed(0)
    shapes = [(32, 64, 256), (32, 224, 256), (16, 128, 512), (16, 480, 512), (8, 256, 1024), (8, 992, 1024),
              (4, 512, 1024), (4, 992, 1024)]
    if args.dense:
        shapes = [(S, K, K) for S, K, _ in shapes]
    if args.only in ('', 'conv1x1'):
        for S, K, ct in shapes:
            M = n * S * S
            A = torch.randn(M, ct, device=DEV)
            W = torch.randn(128, K, device=DEV) * 0.05
            if args.const:
                A.fill_(1.0)
                W.fill_(0.05)
            out = torch.empty(M, 128, device=DEV)
            sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
            scp, shp = (None, None) if args.noact else (L.ptr(sc), L.ptr(sh))
            if args.clamp:
                osc, osh = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
                Wf, bounds, oshf = torch.empty(128, K, device=DEV), torch.empty(2, K, device=DEV), torch.empty(128, device=DEV)
                L.call('gnx_conv1x1_fold_clamp', L.ptr(W), L.ptr(sc), L.ptr(sh), L.ptr(osc), L.ptr(osh), L.ptr(Wf),
                       L.ptr(bounds), L.ptr(oshf), 128, K, st)
                ms = timeit(lambda: L.call('gnx_conv1x1_clamped_act', L.ptr(A), ct, L.ptr(Wf), L.ptr(bounds), L.ptr(out), 128,
                                           M, 128, K, L.ptr(osc), L.ptr(oshf), st), args.reps)
            else:
                ms = timeit(lambda: L.call('gnx_conv1x1_bnrelu', L.ptr(A), ct, L.ptr(W), L.ptr(out), 128, M, 128, K,
                                           scp, shp, 0, 0, st), args.reps)
            fl = 2.0 * M * K * 128
            byts = 4.0 * M * (K + 128)
            print("conv1x1 S=%2d K=%4d M=%8d  %8.3f ms  %6.1f TFLOP/s  %5.2f TB/s" % (S, K, M, ms, fl / ms / 1e9, byts / ms / 1e9))
            del A, out
    if args.only == 'conv1x1split':
        # conv1 + norm2 / relu2 on the store: the fp32 instruction (gnx_conv1x1_bnrelu_act) beside the split-bf16 form
        for S, K, ct in shapes:
            M = n * S * S
            A = torch.randn(M, ct, device=DEV)
            W = torch.randn(128, K, device=DEV) * 0.05
            out = torch.empty(M, 128, device=DEV)
            sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
            osc, osh = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
            Wp = torch.empty(L.query('gnx_conv1x1_split_pack_halves', K), device=DEV, dtype=torch.bfloat16)
            L.call('gnx_conv1x1_split_pack', L.ptr(W), Wp.data_ptr(), K, st)
            ms0 = timeit(lambda: L.call('gnx_conv1x1_bnrelu_act', L.ptr(A), ct, L.ptr(W), L.ptr(out), 128, M, 128, K, L.ptr(sc),
                                        L.ptr(sh), L.ptr(osc), L.ptr(osh), st), args.reps)
            ms1 = timeit(lambda: L.call('gnx_conv1x1_bnrelu_act_split', L.ptr(A), ct, Wp.data_ptr(), L.ptr(out), 128, M, K, L.ptr(sc),
                                        L.ptr(sh), L.ptr(osc), L.ptr(osh), st), args.reps)
            fl = 2.0 * M * K * 128
            byts = 4.0 * M * (K + 128)
            print("conv1x1 S=%2d K=%4d M=%8d  fp32 mfma %8.3f ms %6.1f TFLOP/s %5.2f TB/s | split bf16 %8.3f ms %6.1f TFLOP/s %5.2f TB/s  (x%.2f)" %
                  (S, K, M, ms0, fl / ms0 / 1e9, byts / ms0 / 1e9, ms1, fl / ms1 / 1e9, byts / ms1 / 1e9, ms0 / ms1), flush=True)
            del A, out
    if args.only == 'wgrad1split':
        # conv1's weight gradient: the fp32-instruction kernel (gnx_wgrad_bnrelu) beside the split-bf16 form
        for S, K, ct in shapes:
            M = n * S * S
            X = torch.randn(M, ct, device=DEV)
            dY = torch.randn(M, 128, device=DEV)
            sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
            dW = torch.empty(128, K, device=DEV)
            ws0 = torch.empty(L.query('gnx_wgrad_workspace', M, 128, K, 1), device=DEV)
            ws1 = torch.empty(L.query('gnx_wgrad1x1_split_workspace', M, 128, K), device=DEV)
            ms0 = timeit(lambda: L.call('gnx_wgrad_bnrelu', L.ptr(dY), 128, L.ptr(X), ct, L.ptr(sc), L.ptr(sh), L.ptr(dW), L.ptr(ws0), M, 128,
                                        K, S, 1, 0, 0, st), args.reps)
            ms1 = timeit(lambda: L.call('gnx_wgrad1x1_split', L.ptr(dY), 128, L.ptr(X), ct, L.ptr(sc), L.ptr(sh), L.ptr(dW), L.ptr(ws1), M,
                                        128, K, 0, st), args.reps)
            fl = 2.0 * M * K * 128
            byts = 4.0 * M * (K + 128)
            print("wgrad1x1 S=%2d K=%4d M=%8d  fp32 mfma %8.3f ms %6.1f TFLOP/s %5.2f TB/s | split bf16 %8.3f ms %6.1f TFLOP/s %5.2f TB/s  (x%.2f)" %
                  (S, K, M, ms0, fl / ms0 / 1e9, byts / ms0 / 1e9, ms1, fl / ms1 / 1e9, byts / ms1 / 1e9, ms0 / ms1), flush=True)
            del X, dY
    if args.only == 'wgrad9split':
        # conv2's weight gradient: the fp32-instruction kernel (gnx_wgrad_bnrelu, taps = 9) beside the split-bf16 form
        for S in (32, 16, 8, 4):
            M = n * S * S
            A = torch.relu(torch.randn(M, 128, device=DEV))
            dY = torch.randn(M, 256, device=DEV)
            dW = torch.empty(32, 128, 3, 3, device=DEV)
            ws0 = torch.empty(L.query('gnx_wgrad_workspace', M, 32, 128, 9), device=DEV)
            ws1 = torch.empty(L.query('gnx_wgrad3x3_split_workspace', M), device=DEV)
            ms0 = timeit(lambda: L.call('gnx_wgrad_bnrelu', dY.data_ptr() + 4 * 64, 256, L.ptr(A), 128, None, None, L.ptr(dW), L.ptr(ws0), M, 32,
                                        128, S, 9, 0, 0, st), args.reps)
            ms1 = timeit(lambda: L.call('gnx_wgrad3x3_split', dY.data_ptr() + 4 * 64, 256, L.ptr(A), 128, L.ptr(dW), L.ptr(ws1), M, S, 0, st),
                         args.reps)
            fl = 2.0 * M * 1152 * 32
            byts = 4.0 * M * (128 + 32)
            print("wgrad3x3 S=%2d M=%8d  fp32 mfma %8.3f ms %6.1f TFLOP/s %5.2f TB/s | split bf16 %8.3f ms %6.1f TFLOP/s %5.2f TB/s  (x%.2f)" %
                  (S, M, ms0, fl / ms0 / 1e9, byts / ms0 / 1e9, ms1, fl / ms1 / 1e9, byts / ms1 / 1e9, ms0 / ms1), flush=True)
            del A, dY
    if args.only == 'conv3x3split':
        # conv2 on the ready (activated) bottleneck: Winograd F(2,3) on the fp32 instruction beside the split-bf16 direct form
        for S in (32, 16, 8, 4):
            M = n * S * S
            A = torch.relu(torch.randn(M, 128, device=DEV))
            ct = 256
            out = torch.empty(M, ct, device=DEV)
            Wt = torch.randn(32, 128, 3, 3, device=DEV) * 0.05
            Wu = torch.empty(12, 32, 128, device=DEV)
            L.call('gnx_winograd_conv3x3_weights', L.ptr(Wt), L.ptr(Wu), 32, 128, st)
            Wr = torch.randn(9, 32, 128, device=DEV) * 0.05
            Wp = torch.empty(L.query('gnx_conv3x3_split_pack_halves'), device=DEV, dtype=torch.bfloat16)
            L.call('gnx_conv3x3_split_pack', L.ptr(Wt), Wp.data_ptr(), st)
            if S >= 8:
                ms0 = timeit(lambda: L.call('gnx_conv3x3_winograd', L.ptr(A), 128, L.ptr(Wu), out.data_ptr() + 4 * 64, ct, M, 32, 128, S,
                                            st), args.reps)
            else:
                ms0 = timeit(lambda: L.call('gnx_conv3x3_bnrelu', L.ptr(A), 128, L.ptr(Wr), out.data_ptr() + 4 * 64, ct, M, 32, 128, S,
                                            None, None, st), args.reps)
            ms1 = timeit(lambda: L.call('gnx_conv3x3_split', L.ptr(A), 128, Wp.data_ptr(), out.data_ptr() + 4 * 64, ct, M, S, st), args.reps)
            fl = 2.0 * M * 1152 * 32
            byts = 4.0 * M * (128 + 32)
            print("conv3x3 S=%2d M=%8d  fp32 mfma %8.3f ms %6.1f TFLOP/s (direct-conv FLOPs) %5.2f TB/s | split bf16 %8.3f ms %6.1f TFLOP/s %5.2f TB/s  (x%.2f)" %
                  (S, M, ms0, fl / ms0 / 1e9, byts / ms0 / 1e9, ms1, fl / ms1 / 1e9, byts / ms1 / 1e9, ms0 / ms1), flush=True)
            del A, out
    if args.only in ('', 'conv3x3'):
        for S in (32, 16, 8, 4):
            M = n * S * S
            A = torch.randn(M, 128, device=DEV)
            Wr = torch.randn(9, 32, 128, device=DEV) * 0.05
            ct = 256
            out = torch.empty(M, ct, device=DEV)
            sc, sh = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
            scp, shp = (None, None) if args.noact else (L.ptr(sc), L.ptr(sh))
            if args.wino:
                Wt = torch.randn(32, 128, 3, 3, device=DEV) * 0.05
                Wu = torch.empty(12, 32, 128, device=DEV)
                L.call('gnx_winograd_conv3x3_weights', L.ptr(Wt), L.ptr(Wu), 32, 128, st)
                ms = timeit(lambda: L.call('gnx_conv3x3_winograd', L.ptr(A), 128, L.ptr(Wu), out.data_ptr() + 4 * 64, ct,
                                           M, 32, 128, S, st), args.reps)
            else:
                ms = timeit(lambda: L.call('gnx_conv3x3_bnrelu', L.ptr(A), 128, L.ptr(Wr), out.data_ptr() + 4 * 64, ct,
                                           M, 32, 128, S, scp, shp, st), args.reps)
            fl = 2.0 * M * 1152 * 32
            print("conv3x3 S=%2d M=%8d  %8.3f ms  %6.1f TFLOP/s (direct-conv FLOPs)" % (S, M, ms, fl / ms / 1e9))
            del A, out
    if args.only in ('', 'conv1x1h'):
        # config 5: conv1 on fp16 block buffers
        H = torch.float16
        for S, K, ct in shapes:
            M = n * S * S
            A = torch.randn(M, ct, device=DEV).to(H)
            W = (torch.randn(128, K, device=DEV) * 0.05).to(H)
            out = torch.empty(M, 128, device=DEV, dtype=H)
            sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
            osc, osh = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
            scp, shp = (None, None) if args.noact else (L.ptr(sc), L.ptr(sh))
            ms = timeit(lambda: L.call('gnx_conv1x1_bnrelu_h16', L.ptr(A, H), ct, L.ptr(W, H), L.ptr(out, H), 128, M, 128, K,
                                       scp, shp, L.ptr(osc), L.ptr(osh), st), args.reps)
            print("conv1x1_h16 S=%2d K=%4d M=%8d  %8.3f ms  %6.1f TFLOP/s  %5.2f TB/s (fp16 bytes)" %
                  (S, K, M, ms, 2.0 * M * K * 128 / ms / 1e9, 2.0 * M * (K + 128) / ms / 1e9))
            del A, out
    if args.only in ('fused', 'fusedcmp'):
        # config 5: one dense layer as ONE kernel (gnx_dense_layer_f16) on the 256-px shapes (--spots: images per launch);
        # 'fusedcmp' also times the two-kernel pair it replaces.  TB/s = algorithmic bytes of the fused layer (K columns in,
        # 32 out) over the time.
        H = torch.float16
        L.call('gnx_dense_layer_f16_set_form', args.form)
        fshapes = [(64, 64, 256), (64, 128, 256), (64, 224, 256), (32, 128, 512), (32, 480, 512), (16, 256, 1024),
                   (16, 992, 1024), (8, 512, 1024), (8, 992, 1024)]
        for S, K, ct in fshapes:
            M = n * S * S
            X = torch.randn(ct // 32, M, 32, device=DEV).to(H)          # channel-blocked [ct / 32][rows][32]
            W1 = torch.randn(128, K, device=DEV) * (1.0 / K ** 0.5)
            W2 = torch.randn(32, 128, 3, 3, device=DEV) * 0.05
            sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
            osc, osh = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
            w1p = torch.empty(128 * K, device=DEV, dtype=H)
            w2p = torch.empty(9 * 8 * 512, device=DEV, dtype=H)
            L.call('gnx_dense_layer_f16_pack', L.ptr(W1), L.ptr(W2), L.ptr(w1p, H), L.ptr(w2p, H), K, st)
            ms = timeit(lambda: L.call('gnx_dense_layer_f16', L.ptr(X, H), M, n, S, K, L.ptr(w1p, H), L.ptr(w2p, H), L.ptr(sc),
                                       L.ptr(sh), L.ptr(osc), L.ptr(osh), st), args.reps)
            byts = M * (2.0 * K + 64)
            fl = 2.0 * M * 128 * (K + 9 * 32)
            line = "fused S=%2d K=%4d M=%8d  %8.3f ms  %5.2f TB/s  %6.1f TFLOP/s" % (S, K, M, ms, byts / ms / 1e9, fl / ms / 1e9)
            if args.only == 'fusedcmp' and M * ct < 2 ** 31:
                Xr = torch.randn(M, ct, device=DEV).to(H)                  # the pair works on a row-major buffer
                bott = torch.empty(M, 128, device=DEV, dtype=H)
                W16 = W1.to(H)
                Wr = torch.empty(9, 32, 128, device=DEV)
                L.call('gnx_repack_conv3x3', L.ptr(W2), L.ptr(Wr), 32, 128, st)
                Wr16 = Wr.to(H)

                def pair():
                    L.call('gnx_conv1x1_bnrelu_h16', L.ptr(Xr, H), ct, L.ptr(W16, H), L.ptr(bott, H), 128, M, 128, K, L.ptr(sc),
                           L.ptr(sh), L.ptr(osc), L.ptr(osh), st)
                    L.call('gnx_conv3x3_f16_dma_h', L.ptr(bott, H), 128, L.ptr(Wr16, H), Xr.data_ptr() + 2 * K, ct, M, 32, 128, S, st)
                ms2 = timeit(pair, args.reps)
                line += "   | pair %8.3f ms (x%.2f)" % (ms2, ms2 / ms)
                del bott, Xr
            print(line, flush=True)
            del X
    if args.only in ('', 'wgrad1'):
        # weight gradient of conv1 (taps = 1): dW[128][K] = dY^T act(X); GNX_WGRAD_R1=1 selects the round-1 kernels
        for S, K, ct in shapes:
            M = n * S * S
            X = torch.randn(M, ct, device=DEV)
            dY = torch.randn(M, 128, device=DEV)
            sc, sh = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
            dW = torch.empty(128, K, device=DEV)
            ws = torch.empty(L.query('gnx_wgrad_workspace', M, 128, K, 1), device=DEV)
            ms = timeit(lambda: L.call('gnx_wgrad_bnrelu', L.ptr(dY), 128, L.ptr(X), ct, L.ptr(sc), L.ptr(sh), L.ptr(dW),
                                       L.ptr(ws), M, 128, K, 0, 1, 0, 0, st), args.reps)
            fl = 2.0 * M * K * 128
            print("wgrad1x1 S=%2d K=%4d M=%8d  %8.3f ms  %6.1f TFLOP/s  %5.2f TB/s" %
                  (S, K, M, ms, fl / ms / 1e9, 4.0 * M * (K + 128) / ms / 1e9))
            del X, dY, ws
    if args.only in ('', 'wgrad9'):
        for S in (32, 16, 8, 4):
            M = n * S * S
            X = torch.randn(M, 128, device=DEV)
            ct = 256
            dY = torch.randn(M, ct, device=DEV)
            dW = torch.empty(32, 128, 3, 3, device=DEV)
            sc, sh = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
            scp, shp = (None, None) if args.noact else (L.ptr(sc), L.ptr(sh))
            ws = torch.empty(L.query('gnx_wgrad_workspace', M, 32, 128, 9), device=DEV)
            ms = timeit(lambda: L.call('gnx_wgrad_bnrelu', dY.data_ptr() + 4 * 64, ct, L.ptr(X), 128, scp, shp, L.ptr(dW),
                                       L.ptr(ws), M, 32, 128, S, 9, 0, 0, st), args.reps)
            fl = 2.0 * M * 1152 * 32
            print("wgrad3x3 S=%2d M=%8d  %8.3f ms  %6.1f TFLOP/s" % (S, M, ms, fl / ms / 1e9))
            del X, dY, ws
    if args.only in ('', 'gemm'):
        # the count MLP's first Linear on one array: C[4992][500] = A^T W^T, A = the (genes, spots) count grid read in place
        M, N, K = 4992, 500, 2000
        A = torch.randn(K, M, device=DEV)
        W = torch.randn(N, K, device=DEV) * 0.05
        Wt = W.t().contiguous()
        bias = torch.randn(N, device=DEV)
        C = torch.empty(M, N, device=DEV)
        nws = L.query('gnx_gemm_f32_workspace', M, N, K)
        ws = torch.empty(nws, device=DEV) if nws else None
        print('gemm workspace floats', nws)
        for bk, Bm, ldb in ((0, W, K), (1, Wt, N)):
            ms = timeit(lambda: L.call('gnx_gemm_f32_ws', L.ptr(A), M, 1, L.ptr(Bm), ldb, bk, L.ptr(bias), L.ptr(C), N, M, N, K, 0, L.ptr(ws), st),
                        args.reps)
            print("gemm kmajor-A M=%d N=%d K=%d b_kmajor=%d  %8.3f ms  %6.1f TFLOP/s" % (M, N, K, bk, ms, 2.0 * M * N * K / ms / 1e9))
    if args.only in ('', 'stem'):
        x = torch.rand(n, 3, 128, 128, device=DEV)
        w = torch.randn(64, 3, 7, 7, device=DEV) * 0.05
        out = torch.empty(n * 64 * 64, 64, device=DEV)
        ms = timeit(lambda: L.call('gnx_conv_stem', L.ptr(x), L.ptr(w), L.ptr(out), 64, n, 3, 128, 128, 64, 7, 7, 2, 3,
                                   st), args.reps)
        fl = 2.0 * n * 64 * 64 * 147 * 64
        print("stem7x7 n=%d  %8.3f ms  %6.1f TFLOP/s  (in %.2f GB, out %.2f GB -> %.2f TB/s)" %
              (n, ms, fl / ms / 1e9, x.numel() * 4 / 1e9, out.numel() * 4 / 1e9, (x.numel() + out.numel()) * 4 / ms / 1e9))
        if args.only in ('', 'stem', 'pool'):
            sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1
            pooled = torch.empty(n * 32 * 32, 256, device=DEV)
            ms = timeit(lambda: L.call('gnx_bnrelu_maxpool', L.ptr(out), 64, L.ptr(pooled), 256, n, 64, 64, 64,
                                       L.ptr(sc), L.ptr(sh), st), args.reps)
            print("maxpool n=%d  %8.3f ms  %.2f TB/s" % (n, ms, (out.numel() + n * 32 * 32 * 64) * 4 / ms / 1e9))


if __name__ == '__main__':
    main()
