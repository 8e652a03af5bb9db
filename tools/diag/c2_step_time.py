"""Config 2's step (DenseNet-121 @128 px, batch 32, train-mode BN) timed piece by piece on the device, the way train_spotwise
runs it: zero_grad, the replayed graph (forward + fused CE + backward), the eager optimizer step.
   python tools/diag/c2_step_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gridnext_amd as ga   # noqa: E402
from gridnext_amd import functional as GF, graphs   # noqa: E402

DEV = torch.device('cuda:0')

torch.manual_seed(0)
f = ga.DenseNet(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64,
                bn_size=4, drop_rate=0).to(DEV)
opt = torch.optim.Adam(f.parameters(), lr=1e-3)
x = torch.rand((32, 3, 128, 128), device=DEV)
y = torch.randint(0, 8, (32,), device=DEV)
f.train()


def spot_step(inputs, labels):
    loss, stats, _ = GF.masked_cross_entropy(f(inputs), labels, 1, label_base=0)
    return loss, stats[1], None


stepper = graphs.GridStepGraphs(spot_step, f.parameters(), drop_derived=getattr(f, 'invalidate_cache', None), models=(f,))
for _ in range(6):                                             # eager warm-up batches, then the capture
    opt.zero_grad()
    out = stepper.run(True, x, y)
    if out is None:
        spot_step(x, y)[0].backward()
    opt.step()
assert stepper.run(True, x, y) is not None, "the step was not captured"
torch.cuda.synchronize()
n = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tz = tg = to = 0.0
host_g = host_o = 0.0
for _ in range(n):
    ev[0].record()
    opt.zero_grad()
    ev[1].record()
    h0 = time.perf_counter()
    stepper.run(True, x, y)
    h1 = time.perf_counter()
    ev[2].record()
    opt.step()
    h2 = time.perf_counter()
    ev[3].record()
    torch.cuda.synchronize()
    tz += ev[0].elapsed_time(ev[1])
    tg += ev[1].elapsed_time(ev[2])
    to += ev[2].elapsed_time(ev[3])
    host_g += h1 - h0
    host_o += h2 - h1
print("device time per step: zero_grad %.2f ms | graph replay %.2f ms | optimizer %.2f ms;  host time: replay call %.2f ms, "
      "optimizer call %.2f ms" % (tz / n, tg / n, to / n, 1e3 * host_g / n, 1e3 * host_o / n))
t0 = time.perf_counter()
for _ in range(n):
    opt.zero_grad()
    stepper.run(True, x, y)
    opt.step()
torch.cuda.synchronize()
print("free-running: %.2f ms per step (%.0f spots/s)" % ((time.perf_counter() - t0) / n * 1e3, 32 * n / (time.perf_counter() - t0)))
