#!/usr/bin/env python3
"""Supplementary throughput for the other BASELINE.json configs, driven through the product's own training loops
(bench.py is the headline, C4).  Data are synthetic and resident on the device.

  C1  count-only f (5-Linear MLP, 2000 genes, 8 classes), train_spotwise, batch 128, 19 968 train + 2 560 val spots
  C2  image-only DenseNet-121 f @128 px, train_spotwise (train-mode BN, forward+backward+Adam), batch 32
  C3  count f (frozen) + hex g on 78x64 grids, train_gridwise, 8 train + 2 val arrays, batch 1
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridnext_amd as ga                                   # noqa: E402
from gridnext_amd.synthetic import count_mlp, visium_array   # noqa: E402

DEV = 'cuda:0'


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def c1(epochs):
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 10, (22528, 2000), generator=g).float().to(DEV)
    y = torch.randint(0, 8, (22528,), generator=g).to(DEV)
    dl = {'train': DataLoader(TensorDataset(x[:19968], y[:19968]), batch_size=128, shuffle=True),
          'val': DataLoader(TensorDataset(x[19968:], y[19968:]), batch_size=128)}
    f = count_mlp(2000, 8)
    opt = torch.optim.Adam(f.parameters(), lr=1e-4)
    ga.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=1)          # warm-up
    dt = timed(lambda: ga.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=epochs))
    return {"config": "C1 count-MLP train_spotwise batch 128", "spots_per_s": epochs * 22528 / dt, "seconds": dt}


def c2(n_train, epochs, batch=32):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.rand((n_train + 64, 3, 128, 128), generator=g, device=DEV)
    y = torch.randint(0, 8, (n_train + 64,), device=DEV)
    dl = {'train': DataLoader(TensorDataset(x[:n_train], y[:n_train]), batch_size=batch, shuffle=True),
          'val': DataLoader(TensorDataset(x[n_train:], y[n_train:]), batch_size=batch)}
    f = ga.DenseNet(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6, 12, 24, 16),
                    num_init_features=64, bn_size=4, drop_rate=0)
    opt = torch.optim.Adam(f.parameters(), lr=1e-3)
    ga.train_spotwise(f, {'train': DataLoader(TensorDataset(x[:2 * batch], y[:2 * batch]), batch_size=batch), 'val': dl['val']},
                      nn.CrossEntropyLoss(), opt, num_epochs=1)
    dt = timed(lambda: ga.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=epochs))
    return {"config": "C2 DenseNet-121 @128px train_spotwise batch %d (fwd+bwd+Adam, train-mode BN)" % batch,
            "spots_per_s": epochs * (n_train + 64) / dt, "seconds": dt}


def c3(epochs, fused_adam=False):
    xs, ys = [], []
    for a in range(10):
        _, xc, y = visium_array(a, image=False, device=DEV)
        xs.append(xc)
        ys.append(y)
    x, y = torch.stack(xs), torch.stack(ys)
    dl = {'train': DataLoader(TensorDataset(x[:8], y[:8]), batch_size=1, shuffle=True),
          'val': DataLoader(TensorDataset(x[8:], y[8:]), batch_size=1)}
    m = ga.GridNetHexOddr(count_mlp(2000, 8), (2000,), (78, 64), 8)
    for p in m.patch_classifier.parameters():
        p.requires_grad = False
    # (the tutorials build a plain torch.optim.Adam: its host side - ~170 us per step for g's 14 tensors - is then most of a
    #  count-only step, whose device work is ~250 us; fused_adam = the same optimizer with torch's own fused=True)
    opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3, **({'fused': True} if fused_adam else {}))
    ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=1)
    dt = timed(lambda: ga.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=epochs))
    return {"config": "C3 count f (frozen) + hex g, train_gridwise, 78x64, batch 1" + (", torch.optim.Adam(fused=True)" if fused_adam else ""),
            "spots_per_s": epochs * 10 * 4992 / dt, "arrays_per_s": epochs * 10 / dt, "seconds": dt}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    out = []
    if args.only in ('', 'c1'):
        out.append(c1(3))
    if args.only in ('', 'c3'):
        out.append(c3(40))        # the tutorials train g for 50-100 epochs; 10 would make a fifth of the call warm-up + graph capture
        out.append(c3(40, fused_adam=True))
    if args.only in ('', 'c2'):
        out.append(c2(2048, 1))
    if args.only == 'c2batch':                                # beyond the tutorial's batch of 32: what the kernels do when fed
        for b in (32, 128, 512, 2048):
            out.append(c2(max(2048, 4 * b), 1, b))
    for r in out:
        print(json.dumps(r))
