"""Stage-by-stage comparison of the taped fp16 HIP forward with the oracle's config-5 arithmetic model (fp64 + fp16 rounding points)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn, torch.nn.functional as F
import gridnext_amd as ga
from gridnext_amd import densenet_train_f16 as dt16
from oracle import densenet as odn
DEV = 'cuda:0'
torch.manual_seed(21)
gen = torch.Generator().manual_seed(77)
n = 8
x = torch.rand(n, 3, 128, 128, generator=gen)
m = ga.DenseNet(num_classes=8, **odn.DENSENET121).to(DEV)
bns = [b for b in m.modules() if isinstance(b, nn.BatchNorm2d)]
for b in bns: b.momentum = 1.0
m.train()
with torch.no_grad(): m(x.to(DEV))
m.eval(); m.mfma = 'f16'
class Ctx: pass
ctx = Ctx()
with torch.no_grad():
    out = dt16._DenseNetF16Fn.forward(ctx, m, x.to(DEV), *list(m.parameters()))
tape = ctx.tape
cfg = odn.DenseNetCfg(num_classes=8, **odn.DENSENET121)
sd = {k: v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu() for k, v in m.state_dict().items()}
q = odn.fp16_straight_through
def cmp(name, hip_rows, ref_nchw):
    N, C, S, _ = ref_nchw.shape
    ref = ref_nchw.permute(0, 2, 3, 1).reshape(N * S * S, C)
    hip = hip_rows.double().cpu()[:, :C]
    eq = (hip == ref).double().mean().item()
    err = (hip - ref).abs().max().item() / ref.abs().max().item()
    rms = ((hip - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print('%-48s C %4d  bit-equal %.4f  max err/scale %.2e  rms rel %.2e' % (name, C, eq, err, rms))
feats = None
bi = 0; li = 0
with torch.no_grad():
  for kind, p, ci, co in odn.stages(cfg):
    if kind == 'stem':
        h = F.conv2d(x.double(), sd[p + '.conv0.weight'], None, stride=2, padding=3)
        h = F.relu(odn._bn(sd, p + '.norm0', h, False))
        feats = q(F.max_pool2d(h, 3, 2, 1))
        cmp('stem', tape.bufs[0], feats)
    elif kind == 'dense':
        h = q(F.relu(odn._bn(sd, p + '.norm1', feats, False)))
        h = F.conv2d(h, q(sd[p + '.conv1.weight']))
        h = q(F.relu(odn._bn(sd, p + '.norm2', h, False)))
        a_hip = tape.layers[bi][li][0]
        if li in (0, 1, 5) or li == len(tape.layers[bi]) - 1: cmp(p + ' a', a_hip, h)
        h = q(F.conv2d(h, q(sd[p + '.conv2.weight']), None, padding=1))
        feats = torch.cat([feats, h], 1)
        if li in (0, 1, 5) or li == len(tape.layers[bi]) - 1: cmp(p + ' out', tape.bufs[bi][:, ci:ci + 32], h)
        li += 1
    elif kind == 'transition':
        h = q(F.avg_pool2d(F.relu(odn._bn(sd, p + '.norm', feats, False)), 2, 2))
        cmp(p + ' pooled', tape.trans[bi][1], h)
        feats = q(F.conv2d(h, q(sd[p + '.conv.weight'])))
        bi += 1; li = 0
        cmp(p + ' out', tape.bufs[bi], feats)
