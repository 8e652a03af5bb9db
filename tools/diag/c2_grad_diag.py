"""Diagnostic: which DenseNet-121 parameter gradients differ from the fp64 oracle at C2's shape, and on which inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import gridnext_amd as ga
from oracle import densenet as odn

DEV = 'cuda:0'
cfg = odn.DenseNetCfg(num_classes=8, **odn.DENSENET121)


def run(n, px, training, images):
    labels = torch.tensor([0, 3, 5, 7, 1, 2, 4, 6] * 4)[:n]
    res = {}
    for dtype in (torch.float64, torch.float32):
        sd = odn.closed_form_state(cfg, dtype=dtype)
        x = images(n, px).to(dtype)
        rsd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
               for k, v in sd.items()}
        out = odn.forward(rsd, x, cfg, training=training)
        nn.functional.cross_entropy(out, labels).backward()
        res[dtype] = {k: v.grad.double() for k, v in rsd.items() if v.is_floating_point() and v.grad is not None}
    m = ga.DenseNet(num_classes=8, **odn.DENSENET121)
    m.load_state_dict(odn.closed_form_state(cfg))
    m.to(DEV).train(training)
    out = m(images(n, px).float().to(DEV))
    nn.functional.cross_entropy(out, labels.to(DEV)).backward()
    rows = []
    for k, p in m.named_parameters():
        ref = res[torch.float64][k]
        sc = ref.abs().max().item() + 1e-30
        rows.append(((p.grad.double().cpu() - ref).abs().max().item() / sc,
                     (res[torch.float32][k] - ref).abs().max().item() / sc, k, sc))
    rows.sort(reverse=True)
    print("n=%d px=%d training=%s images=%s" % (n, px, training, images.__name__))
    for r in rows[:6]:
        print("   hip %.3e cpu32 %.3e  %s (|ref|max %.3e)" % r)


def closed(n, px):
    return odn.closed_form_images(n, px)


def rand(n, px):
    return torch.rand(n, 3, px, px, generator=torch.Generator().manual_seed(1))


for args in ((32, 128, True, closed), (32, 128, True, rand), (32, 128, False, closed), (8, 128, True, closed),
             (32, 64, True, closed)):
    run(*args)
