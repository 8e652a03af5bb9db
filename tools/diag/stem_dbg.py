import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch, torch.nn as nn
import test_gpu_bwd_f16 as T
DEV = 'cuda:0'
gen = torch.Generator().manual_seed(79)
x = torch.rand(16, 3, 128, 128, generator=gen).to(DEV)
labels = (torch.arange(16) % 8).to(DEV)
m = T._calibrated_densenet121(23, x)
m.mfma = 'f16'
out = m(x)
nn.functional.cross_entropy(out, labels).backward()
print('flag', int(m.f16_grad_overflow.item()), 'scale', m.f16_grad_scale)
for k, p in m.named_parameters():
    if p.grad is None or not torch.isfinite(p.grad).all():
        print('BAD', k, None if p.grad is None else (~torch.isfinite(p.grad)).sum().item(), p.grad.numel())
print('conv0 grad absmax', m.features.conv0.weight.grad.abs().max().item(), m.features.norm0.weight.grad.abs().max().item())
