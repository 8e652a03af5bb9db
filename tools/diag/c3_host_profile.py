#!/usr/bin/env python3
"""Where the host time of config 3's grid loop goes (cProfile of one train_gridwise call: 8 + 2 arrays x 40 epochs)."""
import cProfile, pstats, io, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset
import gridnext_amd as ga
from gridnext_amd.synthetic import count_mlp, visium_array
DEV = 'cuda:0'
xs, ys = [], []
for a in range(10):
    _, xc, y = visium_array(a, image=False, device=DEV)
    xs.append(xc); ys.append(y)
x, y = torch.stack(xs), torch.stack(ys)
dl = {'train': DataLoader(TensorDataset(x[:8], y[:8]), batch_size=1, shuffle=True), 'val': DataLoader(TensorDataset(x[8:], y[8:]), batch_size=1)}
m = ga.GridNetHexOddr(count_mlp(2000, 8), (2000,), (78, 64), 8)
for p in m.patch_classifier.parameters():
    p.requires_grad = False
opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
with contextlib.redirect_stdout(io.StringIO()):
    ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=40)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
