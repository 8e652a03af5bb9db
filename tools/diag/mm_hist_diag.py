"""Diagnostic: the multimodal tutorial-mode loop histories against the reference fixture, printed with full digits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import contextlib, io
import numpy as np, torch, torch.nn as nn
from torch.utils.data import DataLoader
from conftest import load_golden, sub
import gridnext_amd as ga
from gridnext_amd.synthetic import count_mlp
from test_gpu_models import TINY_LARGE
DEV = 'cuda:0'
g = load_golden('gridwise_hexmm_tutorial')
G, H, W, P, C = 20, 6, 4, 32, 5
m = ga.GridNetHexMM(ga.DenseNet(**TINY_LARGE), count_mlp(G, C), (3, P, P), (G,), (H, W), C)
m.load_state_dict(sub(g, 'init'))
xi, xc, y = torch.from_numpy(g['x_img']), torch.from_numpy(g['x_cnt']), torch.from_numpy(g['y'])
m.to(DEV)
data = [((xi[i], xc[i]), y[i]) for i in range(4)]
dl = {'train': DataLoader(data[:3], batch_size=1, shuffle=False), 'val': DataLoader(data[3:], batch_size=1, shuffle=False)}
opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
for p in m.patch_classifier.parameters():
    p.requires_grad = False
with contextlib.redirect_stdout(io.StringIO()):
    m, vh, th = ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
print('env', os.environ.get('GNX_BN_NO_SMALL'), os.environ.get('GNX_HEX_R1'), os.environ.get('GNX_GRAPH'))
print('train', th, 'ref', list(g['train_history']), 'rel', [abs(a - b) / b for a, b in zip(th, g['train_history'])])
print('val  ', vh, 'ref', list(g['val_history']), 'rel', [abs(a - b) / b for a, b in zip(vh, g['val_history'])])
