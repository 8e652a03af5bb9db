"""Oracle for the count-MLP spot head f_count.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference defines this network in its tutorials from stock torch.nn layers
(notebooks/Tutorial_visium_count.ipynb cell 12, Tutorial_multimodal.ipynb cell 23):
  Linear(G,500) Linear(500,100) BatchNorm1d(100) ReLU
  Linear(100,100) Linear(100,50) BatchNorm1d(50) ReLU  Linear(50,C)
so the oracle is the same stack of stock layers evaluated on the CPU, plus a
functional form that spells the arithmetic out (used to check the stack itself).
"""
import torch
import torch.nn as nn


def count_mlp(n_genes, n_classes):
    return nn.Sequential(
        nn.Linear(n_genes, 500), nn.Linear(500, 100), nn.BatchNorm1d(100), nn.ReLU(),
        nn.Linear(100, 100), nn.Linear(100, 50), nn.BatchNorm1d(50), nn.ReLU(),
        nn.Linear(50, n_classes))


def sequential_forward(seq, x, training):
    """Spelled-out arithmetic of an nn.Sequential of Linear / BatchNorm1d / ReLU.
    Does not touch running statistics; returns (y, [(mean, biased_var, n), ...])."""
    stats = []
    for m in seq:
        if isinstance(m, nn.Linear):
            x = x @ m.weight.t() + m.bias
        elif isinstance(m, nn.BatchNorm1d):
            if training:
                mean = x.mean(0)
                var = x.var(0, unbiased=False)
                stats.append((mean, var, x.shape[0]))
            else:
                mean, var = m.running_mean, m.running_var
            x = (x - mean) / torch.sqrt(var + m.eps) * m.weight + m.bias
        elif isinstance(m, nn.ReLU):
            x = torch.relu(x)
        else:
            raise TypeError(type(m))
    return x, stats
