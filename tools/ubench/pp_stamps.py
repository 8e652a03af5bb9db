import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gridnext_amd import _lib as L
n, S = 4992, 32
M = n * S * S
A = torch.randn(M, 128, device='cuda'); Wr = torch.randn(9, 32, 128, device='cuda') * 0.05
out = torch.empty(M, 256, device='cuda'); sc = torch.rand(128, device='cuda') + 0.5; sh = torch.randn(128, device='cuda') * 0.1
for _ in range(2):
    L.call('gnx_conv3x3_bnrelu', L.ptr(A), 128, L.ptr(Wr), out.data_ptr() + 256, 256, M, 32, 128, S, L.ptr(sc), L.ptr(sh), L.stream())
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8000)()
lib = L.lib(); lib.gnx_debug_pp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
print('rc', lib.gnx_debug_pp_stamps(buf, 8000))
a = np.array(buf[:8000], dtype=np.uint64).reshape(1000, 2, 4)
for half in (0, 1):
    rows = a[10:600, half]
    mem = rows[(rows[:, 3] >> np.uint64(63)) == 1]
    comp = rows[(rows[:, 3] >> np.uint64(63)) == 0]
    t0 = mem[:, 0].astype(np.int64); t1 = mem[:, 1].astype(np.int64); ta = mem[:, 2].astype(np.int64)
    tb = (mem[:, 3] & np.uint64((1 << 63) - 1)).astype(np.int64)
    print('half', half, 'MEM: stash %.0f | fetch issue %.0f | stores+rest %.0f | total %.0f' % (
        (ta - t0).mean(), (tb - ta).mean(), (t1 - tb).mean(), (t1 - t0).mean()))
    c0 = comp[:, 0].astype(np.int64); c1 = comp[:, 1].astype(np.int64)
    print('half', half, 'COMPUTE: body %.0f' % (c1 - c0).mean())
