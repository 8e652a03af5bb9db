#!/usr/bin/env python3
"""Numerical feasibility of SPLIT-OPERAND matrix products for the fp32 path (runs on the CPU, torch only; no product code).

The headline (config 4, fp32) is bound by v_mfma_f32_32x32x2_f32 (157 TFLOP/s dense).  gfx950's 16-bit matrix instructions run at
16x that rate, so an fp32 product computed as THREE 16-bit products with fp32 accumulation,

    a b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi,      a = a_hi + a_lo  (a_hi = a rounded to 16 bits, a_lo = (a - a_hi) rounded),

would run at up to 5x the fp32 instruction's rate.  This script puts a DenseNet-121 (eval-mode BatchNorm, random weights and
statistics as bench.py draws them) through that arithmetic on random patches and reports the deviation of the logits and of the
cross entropy from a float64 evaluation, next to plain fp32, plain fp16 / bf16 operands and the two-term split.

    python tools/diag/split_operand_feasibility.py [patch=64] [n=8]
"""
import sys

import torch
import torch.nn.functional as F

torch.manual_seed(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
G, BN_SIZE, CFG, C0, NC = 32, 4, (6, 12, 24, 16), 64, 8


def split(x, dt, terms):
    hi = x.to(dt).to(torch.float32)
    if terms == 1:
        return [hi]
    lo = (x - hi).to(dt).to(torch.float32)
    return [hi, lo]


def make_mm(mode):
    """mode -> function (a [m,k] fp32, b [k,n] fp32) -> fp32 product with the mode's operand arithmetic, fp32 accumulation"""
    if mode == 'f64':
        return lambda a, b: (a.double() @ b.double())
    if mode == 'f32':
        return lambda a, b: a @ b
    kind, terms = mode.split('x')
    dt = torch.bfloat16 if kind == 'bf16' else torch.float16
    terms = int(terms)

    def mm(a, b):
        A, B = split(a, dt, min(terms, 2)), split(b, dt, min(terms, 2))
        out = A[0] @ B[0]
        if terms >= 2:
            out = out + A[0] @ B[1]
        if terms >= 3:
            out = out + A[1] @ B[0]
        return out
    return mm


def conv(x, w, mm, stride=1, pad=0):
    n, c, h, wd = x.shape
    o, _, kh, kw = w.shape
    cols = F.unfold(x, (kh, kw), padding=pad, stride=stride)            # [n, c kh kw, L]
    ho, wo = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
    a = cols.permute(0, 2, 1).reshape(-1, c * kh * kw)
    y = mm(a.to(torch.float32) if mm is not MM64 else a, w.reshape(o, -1).t().contiguous())
    return y.reshape(n, ho * wo, o).permute(0, 2, 1).reshape(n, o, ho, wo)


def bn(x, p):
    g, b, m, v = p
    return (x - m.view(1, -1, 1, 1)) * (g / torch.sqrt(v + 1e-5)).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def params():
    g = torch.Generator().manual_seed(1)

    def bnp(c):
        return (torch.rand(c, generator=g) * 0.5 + 0.75, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
                torch.rand(c, generator=g) * 0.5 + 0.75)

    def cw(o, i, k):
        return torch.randn(o, i, k, k, generator=g) * (2.0 / (i * k * k)) ** 0.5
    p = {'conv0': cw(C0, 3, 7), 'norm0': bnp(C0), 'blocks': [], 'trans': []}
    c = C0
    for bi, nl in enumerate(CFG):
        layers = []
        for li in range(nl):
            cin = c + li * G
            layers.append((bnp(cin), cw(BN_SIZE * G, cin, 1), bnp(BN_SIZE * G), cw(G, BN_SIZE * G, 3)))
        p['blocks'].append(layers)
        c += nl * G
        if bi + 1 < len(CFG):
            p['trans'].append((bnp(c), cw(c // 2, c, 1)))
            c //= 2
    p['normf'] = bnp(c)
    p['fc'] = (torch.randn(NC, c, generator=g) * (1.0 / c) ** 0.5, torch.zeros(NC))
    return p


def forward(x, p, mm, dt):
    cast = (lambda t: t.double()) if dt == torch.float64 else (lambda t: t)
    T = lambda t: tuple(cast(u) for u in t)                              # noqa: E731
    x = cast(x)
    x = F.max_pool2d(F.relu(bn(conv(x, cast(p['conv0']), mm, 2, 3), T(p['norm0']))), 3, 2, 1)
    for bi, layers in enumerate(p['blocks']):
        for n1, w1, n2, w2 in layers:
            y = conv(F.relu(bn(x, T(n1))), cast(w1), mm)
            y = conv(F.relu(bn(y, T(n2))), cast(w2), mm, 1, 1)
            x = torch.cat([x, y.to(x.dtype)], 1)
        if bi < len(p['trans']):
            nt, wt = p['trans'][bi]
            x = F.avg_pool2d(conv(F.relu(bn(x, T(nt))), cast(wt), mm).to(x.dtype), 2, 2)
    x = F.relu(bn(x, T(p['normf']))).mean((2, 3))
    return x @ cast(p['fc'][0]).t() + cast(p['fc'][1])


MM64 = make_mm('f64')
if __name__ == '__main__':
    p = params()
    x = torch.rand(N, 3, P, P, generator=torch.Generator().manual_seed(2))
    y = torch.randint(0, NC, (N,), generator=torch.Generator().manual_seed(3))
    ref = forward(x, p, MM64, torch.float64)
    ce_ref = F.cross_entropy(ref, y).item()
    scale = ref.abs().max().item()
    print("DenseNet-121, %d patches of %d px: float64 logits max |.| %.3f, CE %.6f" % (N, P, scale, ce_ref))
    for mode in ('f32', 'bf16x3', 'f16x3', 'bf16x2', 'f16x2', 'f16x1', 'bf16x1'):
        out = forward(x, p, make_mm(mode), torch.float32).double()
        d = (out - ref).abs().max().item()
        print("  %-7s max |dlogit| %.3e (%.2e of the range)   |dCE| %.3e   argmax equal %d / %d" %
              (mode, d, d / scale, abs(F.cross_entropy(out, y).item() - ce_ref), int((out.argmax(1) == ref.argmax(1)).sum()), N))
