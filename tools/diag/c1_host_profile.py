#!/usr/bin/env python3
"""Where the host time of config 1's spot loop goes (cProfile of one train_spotwise call over 156 + 20 batches of 128)."""
import cProfile, pstats, io, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset
import gridnext_amd as ga
from gridnext_amd.synthetic import count_mlp
DEV = 'cuda:0'
g = torch.Generator().manual_seed(0)
x = torch.randint(0, 10, (22528, 2000), generator=g).float().to(DEV)
y = torch.randint(0, 8, (22528,), generator=g).to(DEV)
dl = {'train': DataLoader(TensorDataset(x[:19968], y[:19968]), batch_size=128, shuffle=True),
      'val': DataLoader(TensorDataset(x[19968:], y[19968:]), batch_size=128)}
f = count_mlp(2000, 8)
opt = torch.optim.Adam(f.parameters(), lr=1e-4)
with contextlib.redirect_stdout(io.StringIO()):
    ga.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=1)
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    ga.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
