#!/usr/bin/env python3
"""Timers-only stamps of gnx_dense_layer_f16 (diagnostic; runs on the GPU box): builds of csrc/dense_layer_f16.hip with
-DGNX_DL_STAMP -DGNX_DL_NOABL (no run-time ablation branches inside the MFMA streams - with them conv2 reads 6100 cycles per
step instead of 3900) are loaded by name from tools/ubench/build/ and, per shape, the launch time and the cycles per step
wave 0 and wave 4 spent in each stamped segment are printed.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGNX_DL_STAMP -DGNX_DL_NOABL -I gridnext_amd/csrc \
          gridnext_amd/csrc/dense_layer_f16.hip gridnext_amd/csrc/dense_layer_f16_ks.hip -o tools/ubench/build/libdl_noabl.so
    python tools/ubench/dl_timers.py libdl_noabl.so
"""
import ctypes, sys, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
P, I, Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
n = 1664
dev = 'cuda:0'; H = torch.float16
st = torch.cuda.current_stream().cuda_stream
for name in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(HERE, 'build', name))
    lib.gnx_dense_layer_f16_pack.argtypes = [P, P, P, P, I, P]
    lib.gnx_dense_layer_f16.argtypes = [P, Lg, Lg, I, I, P, P, P, P, P, P, P]
    lib.gnx_dense_layer_f16_set_stamps.argtypes = [P, I]
    for S, K, ct in [(64, 64, 256), (64, 224, 256), (32, 128, 512), (32, 480, 512), (16, 992, 1024)]:
        M = n * S * S
        X = torch.randn(ct // 32, M, 32, device=dev).to(H)
        W1 = torch.randn(128, K, device=dev) / K ** 0.5
        W2 = torch.randn(32, 128, 3, 3, device=dev) * 0.05
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        osc, osh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
        w1p, w2p = torch.empty(128 * K, device=dev, dtype=H), torch.empty(9 * 8 * 512, device=dev, dtype=H)
        lib.gnx_dense_layer_f16_pack(W1.data_ptr(), W2.data_ptr(), w1p.data_ptr(), w2p.data_ptr(), K, st)
        def launch():
            lib.gnx_dense_layer_f16(X.data_ptr(), M, n, S, K, w1p.data_ptr(), w2p.data_ptr(), sc.data_ptr(), sh.data_ptr(), osc.data_ptr(), osh.data_ptr(), st)
        lib.gnx_dense_layer_f16_set_stamps(None, 0)
        launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        units = n if S >= 16 else M // 128
        grid = min(units, 256)
        steps = -(-units // grid) * (S * S // 128 if S >= 16 else 1)
        stamps = torch.zeros(grid * 24, device=dev, dtype=torch.int64)
        lib.gnx_dense_layer_f16_set_stamps(stamps.data_ptr(), 0)
        launch(); launch(); torch.cuda.synchronize()
        v = (stamps.view(grid, 24).double().mean(0) / steps).tolist()
        print(name, "S=%d K=%d: %.3f ms  wave0 %s | wave4 %s" % (S, K, ms, " ".join("%.0f" % x for x in v[:8]), " ".join("%.0f" % x for x in v[8:16])), flush=True)
        del X
