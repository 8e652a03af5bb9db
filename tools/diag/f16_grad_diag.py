"""Per-parameter comparison of the fp16 gradient path against the fp32 HIP gradient path (where does an error start?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import gridnext_amd as ga
from oracle import densenet as odn
DEV = 'cuda:0'
torch.manual_seed(21)
gen = torch.Generator().manual_seed(77)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.rand(n, 3, 128, 128, generator=gen).to(DEV)
labels = (torch.arange(n) % 8).to(DEV)
m = ga.DenseNet(num_classes=8, **odn.DENSENET121).to(DEV)
bns = [b for b in m.modules() if isinstance(b, nn.BatchNorm2d)]
for b in bns: b.momentum = 1.0
m.train()
with torch.no_grad(): m(x)
for b in bns: b.momentum = 0.1
m.eval()
res = {}
for mode in ('f32', 'f16'):
    m.mfma = mode
    m.zero_grad()
    out = m(x)
    loss = nn.functional.cross_entropy(out, labels)
    loss.backward()
    res[mode] = {k: p.grad.double().clone() for k, p in m.named_parameters()}
    print(mode, 'loss', loss.item())
print('scale', m.f16_grad_scale.tolist(), 'flag', m.f16_grad_overflow.item())
names = list(res['f32'])
fa = torch.cat([res['f16'][k].reshape(-1) for k in names]); fb = torch.cat([res['f32'][k].reshape(-1) for k in names])
print('N', n, 'TOTAL cos', (fa @ fb / (fa.norm() * fb.norm())).item())
if len(sys.argv) > 2: sys.exit(0)
for k in names[::-1]:
    a, b = res['f16'][k].reshape(-1), res['f32'][k].reshape(-1)
    cos = (a @ b / (a.norm() * b.norm() + 1e-300)).item()
    ratio = (a.norm() / (b.norm() + 1e-300)).item()
    if 'denselayer' in k and not any(t in k for t in ('denselayer1.', 'denselayer16.', 'denselayer24.', 'denselayer12.', 'denselayer6.')):
        continue
    print('%-55s cos %.5f  |f16|/|f32| %.4f  n %d' % (k, cos, ratio, a.numel()))
