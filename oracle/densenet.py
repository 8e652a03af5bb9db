"""Oracle for the DenseNet-BC image classifier f.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A functional restatement of gridnext/densenet.py over a flat ``state`` dict that
uses the reference's state_dict key names (so reference checkpoints, the
fixtures and the HIP product all exchange weights by name):

  stem        densenet.py:98-112   conv0 [7x7 s2 p3 | 3x3 s1 p1] (+ norm0, relu0, pool0 3x3 s2 p1)
  dense layer densenet.py:21-44    cat -> norm1 -> relu -> conv1 1x1 -> norm2 -> relu -> conv2 3x3 p1 [-> dropout]
  dense block densenet.py:57-75    features appended, final cat
  transition  densenet.py:47-54    norm -> relu -> conv 1x1 -> avgpool 2x2
  tail        densenet.py:136-159  norm_final -> relu -> adaptive_avg_pool(1,1) -> flatten [-> classifier]
  init        densenet.py:141-150

Pinned by tests/golden/densenet_tiny_*.npz and densenet121_closedform.npz, which
tools/gen_golden.py produced by running the reference's own DenseNet class.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass
class DenseNetCfg:
    growth_rate: int = 12
    block_config: Tuple[int, ...] = (16, 16, 16)
    compression: float = 0.5
    num_init_features: int = 24
    bn_size: int = 4
    drop_rate: float = 0.0
    num_classes: int = 10
    small_inputs: bool = True
    classify: bool = True


DENSENET121 = dict(growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64,
                   bn_size=4, drop_rate=0, small_inputs=False)


def stages(cfg):
    """Ordered description of the network: list of (kind, prefix, c_in, c_out)."""
    out = [('stem', 'features', 3, cfg.num_init_features)]
    c = cfg.num_init_features
    for bi, n_layers in enumerate(cfg.block_config):
        for li in range(n_layers):
            out.append(('dense', 'features.denseblock%d.denselayer%d' % (bi + 1, li + 1),
                        c + li * cfg.growth_rate, cfg.growth_rate))
        c = c + n_layers * cfg.growth_rate
        if bi != len(cfg.block_config) - 1:
            c_out = int(c * cfg.compression)
            out.append(('transition', 'features.transition%d' % (bi + 1), c, c_out))
            c = c_out
    out.append(('tail', 'features.norm_final', c, c))
    out.append(('classifier', 'classifier', c, cfg.num_classes))
    return out


def _bn_entries(prefix, c):
    return [(prefix + '.weight', (c,)), (prefix + '.bias', (c,)), (prefix + '.running_mean', (c,)),
            (prefix + '.running_var', (c,)), (prefix + '.num_batches_tracked', ())]


def state_layout(cfg):
    """[(key, shape)] in the order the reference's state_dict() lists them."""
    ent = []
    for kind, p, ci, co in stages(cfg):
        if kind == 'stem':
            k = 3 if cfg.small_inputs else 7
            ent.append((p + '.conv0.weight', (co, 3, k, k)))
            if not cfg.small_inputs:
                ent += _bn_entries(p + '.norm0', co)
        elif kind == 'dense':
            mid = cfg.bn_size * cfg.growth_rate
            ent += _bn_entries(p + '.norm1', ci)
            ent.append((p + '.conv1.weight', (mid, ci, 1, 1)))
            ent += _bn_entries(p + '.norm2', mid)
            ent.append((p + '.conv2.weight', (co, mid, 3, 3)))
        elif kind == 'transition':
            ent += _bn_entries(p + '.norm', ci)
            ent.append((p + '.conv.weight', (co, ci, 1, 1)))
        elif kind == 'tail':
            ent += _bn_entries(p, ci)
        else:
            ent.append((p + '.weight', (co, ci)))
            ent.append((p + '.bias', (co,)))
    return ent


def init_state(cfg, generator=None, dtype=torch.float32):
    """Random init with the reference's distributions (densenet.py:141-150)."""
    sd = OrderedDict()
    for key, shape in state_layout(cfg):
        if key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif key.endswith('running_mean'):
            sd[key] = torch.zeros(shape, dtype=dtype)
        elif key.endswith('running_var'):
            sd[key] = torch.ones(shape, dtype=dtype)
        elif 'conv' in key:
            n = shape[0] * shape[2] * shape[3]
            sd[key] = torch.randn(shape, generator=generator, dtype=dtype) * math.sqrt(2.0 / n)
        elif 'norm' in key:
            sd[key] = torch.ones(shape, dtype=dtype) if key.endswith('weight') else torch.zeros(shape, dtype=dtype)
        elif key == 'classifier.weight':
            bound = 1.0 / math.sqrt(shape[1])       # nn.Linear default (kaiming_uniform a=sqrt(5))
            sd[key] = (torch.rand(shape, generator=generator, dtype=dtype) * 2 - 1) * bound
        else:
            sd[key] = torch.zeros(shape, dtype=dtype)
    return sd


def closed_form_state(cfg, dtype=torch.float32):
    """Deterministic fill both the fixture generator and the tests can rebuild
    without shipping a 28 MB blob: element i of entry e gets a scaled sine."""
    sd = OrderedDict()
    for e, (key, shape) in enumerate(state_layout(cfg)):
        n = 1
        for s in shape:
            n *= s
        i = torch.arange(n, dtype=torch.float64)
        wave = torch.sin(i * 0.618033988749895 + 0.37 * e)
        if key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif key.endswith('running_var'):
            sd[key] = (1.0 + 0.5 * wave * wave).to(dtype).reshape(shape)
        elif key.endswith('running_mean'):
            sd[key] = (0.1 * wave).to(dtype).reshape(shape)
        elif 'norm' in key and key.endswith('weight'):
            sd[key] = (1.0 + 0.25 * wave).to(dtype).reshape(shape)
        elif 'norm' in key and key.endswith('bias'):
            sd[key] = (0.1 * wave).to(dtype).reshape(shape)
        elif 'conv' in key:
            fan = shape[1] * shape[2] * shape[3]
            sd[key] = (wave * math.sqrt(3.0 / fan)).to(dtype).reshape(shape)
        elif key == 'classifier.weight':
            sd[key] = (wave * math.sqrt(3.0 / shape[1])).to(dtype).reshape(shape)
        else:
            sd[key] = (0.05 * wave).to(dtype).reshape(shape)
    return sd


def closed_form_images(n, p, dtype=torch.float32):
    """Deterministic (n, 3, p, p) images in [0, 1)."""
    i = torch.arange(n * 3 * p * p, dtype=torch.float64)
    v = 0.5 + 0.5 * torch.sin(i * 0.7548776662466927 + 0.1)
    return (v * 0.999).to(dtype).reshape(n, 3, p, p)


def _bn(sd, prefix, x, training, momentum=0.1, eps=1e-5):
    y = F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                     sd[prefix + '.weight'], sd[prefix + '.bias'], training, momentum, eps)
    if training:
        sd[prefix + '.num_batches_tracked'] += 1
    return y


def fp16_straight_through(t):
    """Round to IEEE fp16 in the forward, identity in the backward: the arithmetic model of an fp16 STORAGE / OPERAND point."""
    return t + (t.detach().to(torch.float16).to(t.dtype) - t.detach())


def forward(sd, x, cfg, training=False, return_features=False, dropout_masks=None, quant=None, tap=None):
    """DenseNet.forward (densenet.py:152-159). `sd` BN buffers are updated in place when training.
    dropout_masks: optional iterator of keep-masks (N, growth, S, S), one per dense layer in order, used INSTEAD of
    F.dropout's own random mask (densenet.py:42-43: new_features * mask / (1 - p)) so a test can give both sides the same one.
    quant: None = the reference's arithmetic.  A callable (`fp16_straight_through`) = the arithmetic MODEL of BASELINE config
    5's "fp16 MFMA conv path" as this build defines it (the reference has no reduced-precision path; under torch.autocast the
    same tensors would be halves): the concatenated features, every conv's two matrix operands (activated input, weight) and
    the transition's pooled operand are rounded to fp16 where the HIP path stores / stages them; BatchNorm + ReLU, every
    accumulation, the stem, the final pool and the classifier stay in the working precision; transitions pool first (the
    HIP path's order; equal in exact arithmetic).  Gradients pass the rounding points unchanged.
    tap: optional callable (name, tensor) -> tensor applied to every STORED tensor of that model ('stem', '<layer>.a' = the
    activated bottleneck, '<layer>.out' = the layer's new features, '<transition>.pooled', '<transition>.out'): a test can
    record them, or substitute another implementation's values (straight-through) so that both sides differentiate at the
    same activations."""
    feats = None
    if quant is not None and tap is not None:
        q = lambda t, name=None: tap(name, quant(t)) if name else quant(t)       # noqa: E731
    elif quant is not None:
        q = lambda t, name=None: quant(t)                                        # noqa: E731
    else:
        q = lambda t, name=None: t                                               # noqa: E731
    masks = iter(dropout_masks) if dropout_masks is not None else None
    for kind, p, ci, co in stages(cfg):
        if kind == 'stem':
            if cfg.small_inputs:
                feats = F.conv2d(x, sd[p + '.conv0.weight'], None, stride=1, padding=1)
            else:
                h = F.conv2d(x, sd[p + '.conv0.weight'], None, stride=2, padding=3)
                h = F.relu(_bn(sd, p + '.norm0', h, training))
                feats = q(F.max_pool2d(h, kernel_size=3, stride=2, padding=1), 'stem')
        elif kind == 'dense':
            h = q(F.relu(_bn(sd, p + '.norm1', feats, training)))
            h = F.conv2d(h, q(sd[p + '.conv1.weight']))
            h = q(F.relu(_bn(sd, p + '.norm2', h, training)), p + '.a')
            h = q(F.conv2d(h, q(sd[p + '.conv2.weight']), None, padding=1), p + '.out')
            if cfg.drop_rate > 0 and training and masks is not None:
                h = h * next(masks).to(h.dtype) / (1.0 - cfg.drop_rate)
            elif cfg.drop_rate > 0:
                h = F.dropout(h, p=cfg.drop_rate, training=training)
            feats = torch.cat([feats, h], dim=1)
        elif kind == 'transition' and quant is not None:
            h = q(F.avg_pool2d(F.relu(_bn(sd, p + '.norm', feats, training)), kernel_size=2, stride=2), p + '.pooled')
            feats = q(F.conv2d(h, q(sd[p + '.conv.weight'])), p + '.out')
        elif kind == 'transition':
            h = F.relu(_bn(sd, p + '.norm', feats, training))
            h = F.conv2d(h, sd[p + '.conv.weight'])
            feats = F.avg_pool2d(h, kernel_size=2, stride=2)
        elif kind == 'tail':
            h = F.relu(_bn(sd, p, feats, training))
            feats = torch.flatten(F.adaptive_avg_pool2d(h, (1, 1)), 1)
        elif kind == 'classifier' and cfg.classify:
            feats = F.linear(feats, sd[p + '.weight'], sd[p + '.bias'])
    return feats


class DenseNet(torch.nn.Module):
    """nn.Module shell around the functional oracle (for loops/optimizers).
    Parameters and buffers are registered under the reference's key names."""

    def __init__(self, growth_rate=12, block_config=(16, 16, 16), compression=0.5,
                 num_init_features=24, bn_size=4, drop_rate=0, num_classes=10,
                 small_inputs=True, efficient=False, classify=True, generator=None):
        super().__init__()
        self.cfg = DenseNetCfg(growth_rate, tuple(block_config), compression, num_init_features,
                               bn_size, drop_rate, num_classes, small_inputs, classify)
        self._keys = []
        for key, val in init_state(self.cfg, generator).items():
            name = key.replace('.', '__')
            self._keys.append((key, name))
            if key.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
                self.register_buffer(name, val)
            else:
                self.register_parameter(name, torch.nn.Parameter(val))

    def named_state(self):
        return OrderedDict((k, getattr(self, n)) for k, n in self._keys)

    def load_named_state(self, sd):
        with torch.no_grad():
            for k, n in self._keys:
                getattr(self, n).copy_(torch.as_tensor(sd[k]))

    def forward(self, x):
        return forward(self.named_state(), x, self.cfg, self.training)
