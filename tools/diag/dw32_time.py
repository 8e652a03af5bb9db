"""Sustained rate of gnx_conv1x1_dgrad_wgrad_bnrelu_bwd at the shapes of a 128-px array (4992 spots)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gridnext_amd import _lib as L
DEV = 'cuda:0'
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for S, cins in ((32, (64, 96, 128, 160, 224)), (16, (128, 256, 288, 480)), (8, (256, 512, 992)), (4, (512, 992))):
    M = 4992 * S * S
    for cin in cins:
        ld = {32: 256, 16: 512, 8: 1024, 4: 1024}[S]
        dB = torch.randn(M, 128, device=DEV)
        X = torch.randn(M, ld, device=DEV)
        G = torch.randn(M, ld, device=DEV)
        Wt = torch.randn(cin, 128, device=DEV) * 0.1
        sc, sh, mu, inv = (torch.rand(cin, device=DEV) + 0.5 for _ in range(4))
        dg, db, dW = torch.empty(cin, device=DEV), torch.empty(cin, device=DEV), torch.empty(128, cin, device=DEV)
        ws = torch.empty(L.query('gnx_conv1x1_dgrad_wgrad_workspace', M, cin), device=DEV)
        def run():
            L.call('gnx_conv1x1_dgrad_wgrad_bnrelu_bwd', L.ptr(dB), 128, L.ptr(Wt), L.ptr(X), ld, L.ptr(G), ld, M, cin, L.ptr(sc),
                   L.ptr(sh), L.ptr(mu), L.ptr(inv), L.ptr(dg), L.ptr(db), L.ptr(dW), L.ptr(ws), 0, L.stream())
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        fl = 4.0 * M * cin * 128
        by = 4.0 * M * (128 + 3 * cin)
        print("S %2d cin %4d: %.3f ms  %.1f TFLOP/s  %.2f TB/s (algorithmic)" % (S, cin, dt * 1e3, fl / dt / 1e12, by / dt / 1e12))
        del dB, X, G
