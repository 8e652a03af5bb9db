"""Config 2 (DenseNet-121 @128 px, train_spotwise, batch 32) with groups of parameters frozen: how much of the step is the
weight-gradient branch (conv weights frozen: no weight-gradient kernels or their reductions), the BatchNorm parameter
gradients, the optimizer.  Upper bounds for what optimising each could give; results are NOT training runs."""
import os
import sys

import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gridnext_amd as ga   # noqa: E402

DEV = 'cuda:0'


def run(freeze, batch=32, n_train=2048):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.rand((n_train + 64, 3, 128, 128), generator=g, device=DEV)
    y = torch.randint(0, 8, (n_train + 64,), device=DEV)
    dl = {'train': DataLoader(TensorDataset(x[:n_train], y[:n_train]), batch_size=batch, shuffle=True),
          'val': DataLoader(TensorDataset(x[n_train:], y[n_train:]), batch_size=batch)}
    f = ga.DenseNet(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64,
                    bn_size=4, drop_rate=0)
    for name, p in f.named_parameters():
        if ('conv' in name and 'conv' in freeze) or ('norm' in name and 'norm' in freeze):
            p.requires_grad = False
    params = [p for p in f.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    import contextlib, io, time
    with contextlib.redirect_stdout(io.StringIO()):
        ga.train_spotwise(f, {'train': DataLoader(TensorDataset(x[:2 * batch], y[:2 * batch]), batch_size=batch), 'val': dl['val']},
                          nn.CrossEntropyLoss(), opt, num_epochs=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ga.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    steps = n_train // batch
    print("frozen %-12s: %.0f spots/s, %.2f ms per train step (val included in the time)" % (freeze or '-', (n_train + 64) / dt, 1e3 * dt / steps))


for fz in ('', 'conv', 'norm', 'conv+norm'):
    run(fz)
