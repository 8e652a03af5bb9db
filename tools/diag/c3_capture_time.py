"""What one call of train_gridwise pays before its step graphs replay (config 3: count f frozen + hex g, one 78 x 64 array): the
eager warm-up steps and the capture, per (phase) key, wall-clock with the device drained before and after each.
python tools/diag/c3_capture_time.py"""
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
for call in range(3):                       # three "loop calls": a fresh stepper each, as train_gridwise builds one per call
    stepper = graphs.GridStepGraphs(lambda i, l: gtrain._grid_loss(m, i, l, crit, 1, True), m.parameters(), models=(m,))
    line = []
    for k in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = stepper.run(True, x, y)
        if out is None:
            gtrain._grid_loss(m, x, y, crit, 1, True)[0].backward()
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        line.append("%s %.2f ms" % ("eager" if out is None else "graph", 1e3 * (time.perf_counter() - t0)))
    print("call %d: %s" % (call, " | ".join(line)), flush=True)
    with torch.no_grad():
        line = []
        m.eval()
        for k in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = stepper.run(False, x, y)
            if out is None:
                gtrain._grid_loss(m, x, y, crit, 1, True)
            torch.cuda.synchronize()
            line.append("%s %.2f ms" % ("eager" if out is None else "graph", 1e3 * (time.perf_counter() - t0)))
        m.train()
        m.patch_classifier.eval()
    print("   val: %s" % " | ".join(line), flush=True)
