"""Sustained rate of gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16 (conv1 data + weight gradient, one pass) at the layer shapes of a
256-px array of `spots` spots (default 4992: config 5)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
spots = int(sys.argv[1]) if len(sys.argv) > 1 else 4992
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # one map size only (PMC passes)
ls = torch.tensor([1.0, 1.0], device=DEV)
flag = torch.zeros(1, dtype=torch.int32, device=DEV)
tot = 0.0
for S, ld, cins, n_layers in ((64, 256, (64, 128, 160, 224), 6), (32, 512, (128, 256, 384, 480), 12), (16, 1024, (256, 512, 768, 992), 24),
                              (8, 1024, (512, 768, 992), 16)):
    if only and S != only:
        continue
    M = spots * S * S
    dB = (torch.randn(M, 128, device=DEV) * 0.1).half()
    X = torch.randn(M, ld, device=DEV, dtype=torch.float16)
    G = (torch.randn(M, ld, device=DEV) * 0.1).half()
    for cin in cins:
        Wt = (torch.randn(cin, 128, device=DEV) * 0.05).half()
        sc, sh, mu, inv = (torch.rand(cin, device=DEV) + 0.5 for _ in range(4))
        sh = sh - 1.0
        dg, db, dW = torch.empty(cin, device=DEV), torch.empty(cin, device=DEV), torch.empty(128, cin, device=DEV)
        ws = torch.empty(L.query('gnx_conv1x1_dgrad_wgrad_f16_workspace', M, cin), device=DEV)
        def run():
            L.call('gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16', dB.data_ptr(), Wt.data_ptr(), X.data_ptr(), ld, G.data_ptr(), ld, M, cin,
                   L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(inv), L.ptr(dg), L.ptr(db), L.ptr(dW), L.ptr(ws), L.ptr(ls), 0,
                   flag.data_ptr(), L.stream())
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        by = 2.0 * M * (128 + 3 * cin)
        print("S %2d cin %4d: %7.3f ms  %5.1f TFLOP/s  %.2f TB/s (algorithmic)" % (S, cin, dt * 1e3, 4.0 * M * cin * 128 / dt / 1e12, by / dt / 1e12), flush=True)
    del dB, X, G
