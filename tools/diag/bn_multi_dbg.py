import sys, torch, torch.nn as nn
sys.path.insert(0, '.')
from gridnext_amd import functional as GF, _lib as L
DEV = 'cuda:0'
def run(M, C, tag):
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 3 + 5)
    dy = torch.randn(M, C, generator=g)
    bn = nn.BatchNorm1d(C)
    ref = nn.BatchNorm1d(C)
    ref.load_state_dict(bn.state_dict())
    bn = bn.to(DEV).train(); ref.train()
    xr = x.clone().requires_grad_(True); yr = torch.relu(ref(xr)); yr.backward(dy)
    xd = x.to(DEV).requires_grad_(True); yd = GF.batch_norm_relu(xd, bn, True); yd.backward(dy.to(DEV))
    torch.cuda.synchronize()
    e = lambda a, b: float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max())
    print(tag, M, C, 'y %.2e dx %.2e dgamma %.2e dbeta %.2e' % (e(yd, yr), e(xd.grad, xr.grad), e(bn.weight.grad, ref.weight.grad), e(bn.bias.grad, ref.bias.grad)),
          'sync', bn._gnx_sync.tolist()[:12] if getattr(bn, '_gnx_sync', None) is not None else None, flush=True)
for M, C in ((4992, 32), (8192, 512), (5000, 1024)):
    run(M, C, 'fresh')
# disturb: what the suite does before (a big fp16 workload, other kernels)
import gridnext_amd as ga
m = ga.DenseNet(growth_rate=32, block_config=(2, 2), num_init_features=64, bn_size=4, num_classes=5, small_inputs=False).to(DEV)
m.train()
out = m(torch.rand(32, 3, 128, 128, device=DEV)); out.sum().backward()
for M, C in ((4992, 32), (8192, 512), (5000, 1024)):
    run(M, C, 'after densenet train step')
