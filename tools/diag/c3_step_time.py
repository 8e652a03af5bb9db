"""Config 3's step (count f frozen + hex g on one 78 x 64 array, train_gridwise) timed piece by piece on the device, the way the
loop runs it: the replayed graph (f forward, g forward, masked CE, g backward), the eager optimizer step + zero_grad, and the
host time of each.   python tools/diag/c3_step_time.py"""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gridnext_amd as ga   # noqa: E402
from gridnext_amd import graphs, training as gtrain   # noqa: E402
from gridnext_amd.synthetic import count_mlp, visium_array   # noqa: E402

DEV = torch.device('cuda:0')
torch.manual_seed(0)
m = ga.GridNetHexOddr(count_mlp(2000, 8), (2000,), (78, 64), 8).to(DEV)
for p in m.patch_classifier.parameters():
    p.requires_grad = False
opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
crit = nn.CrossEntropyLoss()
_, xc, y = visium_array(3, image=False, device=DEV)
x, y = xc.unsqueeze(0), y.unsqueeze(0)
m.train()
m.patch_classifier.eval()
stepper = graphs.GridStepGraphs(lambda i, l: gtrain._grid_loss(m, i, l, crit, 1, True), m.parameters(), models=(m,))
for _ in range(6):
    out = stepper.run(True, x, y)
    if out is None:
        gtrain._grid_loss(m, x, y, crit, 1, True)[0].backward()
    opt.step()
    opt.zero_grad()
assert stepper.run(True, x, y) is not None, "the step was not captured"
opt.step(); opt.zero_grad()
torch.cuda.synchronize()
n = 200
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tg = to = hg = ho = 0.0
for _ in range(n):
    ev[0].record()
    h0 = time.perf_counter()
    stepper.run(True, x, y)
    h1 = time.perf_counter()
    ev[1].record()
    opt.step()
    opt.zero_grad()
    h2 = time.perf_counter()
    ev[2].record()
    torch.cuda.synchronize()
    tg += ev[0].elapsed_time(ev[1]); to += ev[1].elapsed_time(ev[2]); hg += h1 - h0; ho += h2 - h1
print("device time per array: graph replay %.1f us | optimizer + zero_grad %.1f us;  host: replay call %.1f us, optimizer %.1f us" %
      (1e3 * tg / n, 1e3 * to / n, 1e6 * hg / n, 1e6 * ho / n))
t0 = time.perf_counter()
for _ in range(n):
    stepper.run(True, x, y)
    opt.step()
    opt.zero_grad()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("free-running: %.1f us per array (%.2f M spots/s)" % (1e6 * dt, 4992 / dt / 1e6))
