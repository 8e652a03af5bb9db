"""Diagnostic: BASELINE config 3 (count f frozen + hex g, train_gridwise, batch 1) - what a loop call costs with and
without hipGraph replay, and how much of a short call is capture."""
import contextlib, io, os, sys, time
import torch, torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gridnext_amd as ga
from gridnext_amd.synthetic import count_mlp, visium_array
DEV = 'cuda:0'
xs, ys = [], []
for a in range(10):
    _, xc, y = visium_array(a, image=False, device=DEV)
    xs.append(xc); ys.append(y)
x, y = torch.stack(xs), torch.stack(ys)
dl = {'train': DataLoader(TensorDataset(x[:8], y[:8]), batch_size=1, shuffle=True),
      'val': DataLoader(TensorDataset(x[8:], y[8:]), batch_size=1)}
m = ga.GridNetHexOddr(count_mlp(2000, 8), (2000,), (78, 64), 8)
for p in m.patch_classifier.parameters():
    p.requires_grad = False
opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
def run(epochs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=epochs)
    torch.cuda.synchronize(); return time.perf_counter() - t0
flags = sys.argv[1].split(',') if len(sys.argv) > 1 else ('1', '0')
eps = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 10, 40)
for flag in flags:
    os.environ['GNX_GRAPH'] = flag
    run(1)
    for ep in eps:
        dt = run(ep)
        print("GNX_GRAPH=%s epochs %3d: %.1f ms, %.3f ms/array, %.2f M spots/s" % (flag, ep, dt * 1e3, dt * 1e3 / (10 * ep), ep * 49920 / dt / 1e6), flush=True)
