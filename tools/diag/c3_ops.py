"""Which host-side torch ops launch the fill / copy kernels of a config-3 step (count f frozen + hex g, train_gridwise)?
Eager (GNX_GRAPH=0) and under torch.profiler with stacks: prints aten ops that launch device kernels, by call site."""
import os, sys, collections
os.environ['GNX_GRAPH'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset
import gridnext_amd as ga
from gridnext_amd.synthetic import count_mlp, visium_array
DEV = 'cuda:0'
xs, ys = [], []
for a in range(4):
    _, xc, y = visium_array(a, image=False, device=DEV)
    xs.append(xc); ys.append(y)
x, y = torch.stack(xs), torch.stack(ys)
dl = {'train': DataLoader(TensorDataset(x[:3], y[:3]), batch_size=1, shuffle=True), 'val': DataLoader(TensorDataset(x[3:], y[3:]), batch_size=1)}
m = ga.GridNetHexOddr(count_mlp(2000, 8), (2000,), (78, 64), 8)
for p in m.patch_classifier.parameters():
    p.requires_grad = False
opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
import contextlib, io, traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ('fill_', 'zero_', 'zeros', 'ones_like', 'copy_', 'full', 'clone', '_to_copy')):
            st = [f for f in traceback.extract_stack() if 'c3_ops' not in f.filename and 'python_dispatch' not in f.filename][-3:]
            agg[(name, tuple('%s:%d' % (f.filename.split('/')[-1], f.lineno) for f in st))] += 1
        return func(*args, **(kwargs or {}))
with contextlib.redirect_stdout(io.StringIO()):
    ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=1)
    with Log():
        ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
for (n, st), c in agg.most_common(40):
    print(c, n, ' <- '.join(st))
