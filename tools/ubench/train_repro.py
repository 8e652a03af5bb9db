"""Debug: training forward/backward (eval-mode BN) with and without the Winograd conv2, with a dirty allocator."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import gridnext_amd as ga
from oracle import densenet as odn
DEV = 'cuda:0'
if len(sys.argv) > 1:
    junk = [torch.full((sz,), float('nan'), device=DEV) for sz in (10, 100, 1000, 10**4, 10**5, 10**6, 10**7, 3 * 10**7) for _ in range(8)]
    del junk
cfg = odn.DenseNetCfg(num_classes=8, **odn.DENSENET121)
labels = torch.tensor([0, 3, 5, 7, 1, 2]).to(DEV)
m = ga.DenseNet(num_classes=8, **odn.DENSENET121)
m.load_state_dict(odn.closed_form_state(cfg))
m.to(DEV).train(False)
x = odn.closed_form_images(6, 64).to(DEV)
res = {}
for wino in (True, False, True):
    m.winograd = wino
    m.zero_grad(set_to_none=True)
    out = m(x)
    loss = nn.functional.cross_entropy(out, labels)
    loss.backward()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    print('wino', wino, 'loss', loss.item(), 'nan grads', sum(int(torch.isnan(v).any()) for v in g.values()))
    res.setdefault(wino, []).append((out.detach().clone(), g))
a, b = res[True][0], res[False][0]
print('wino twice identical', all(torch.equal(res[True][0][1][k], res[True][1][1][k]) for k in a[1]))
worst = sorted(((((a[1][k] - b[1][k]).abs().max() / (b[1][k].abs().max() + 1e-30)).item(), k) for k in a[1]), reverse=True)[:6]
print('largest relative grad differences wino vs direct:', worst)
