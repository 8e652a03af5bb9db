"""Diagnostic: gradients of the f-trained (eval-statistics) DenseNet-121 step, Winograd vs direct conv2, and run-to-run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn as nn
import gridnext_amd as ga
DEV = 'cuda:0'
torch.manual_seed(11)
m = ga.DenseNet(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64,
                bn_size=4, drop_rate=0).to(DEV)
x = torch.rand(8, 3, 64, 64, device=DEV)
for b in m.modules():
    if isinstance(b, nn.BatchNorm2d):
        b.momentum = 1.0
m.train()
with torch.no_grad():
    m(x)
m.eval()
labels = torch.arange(8, device=DEV) % 8
def run(wino):
    m.winograd = wino
    m.zero_grad()
    out = m(x)
    loss = nn.functional.cross_entropy(out, labels)
    loss.backward()
    return out.detach().clone(), loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}
a, b, c = run(False), run(False), run(True)
def cmp(u, v, what):
    errs = np.array([((v[2][k] - g).abs().max() / (g.abs().max() + 1e-30)).item() for k, g in u[2].items()])
    names = list(u[2].keys())
    worst = np.argsort(-errs)[:5]
    print(what, 'logits diff %.3e loss diff %.3e | median %.3e q90 %.3e max %.3e' % (
        (u[0] - v[0]).abs().max().item(), abs(u[1] - v[1]), np.median(errs), np.quantile(errs, 0.9), errs.max()),
        [(names[i], '%.2e' % errs[i]) for i in worst])
print('env GNX_BN_NO_SMALL =', os.environ.get('GNX_BN_NO_SMALL'))
cmp(a, b, 'direct vs direct  :')
cmp(a, c, 'direct vs winograd:')
