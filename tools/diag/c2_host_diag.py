"""Diagnostic: is a batch-32 DenseNet-121 training step bound by the host (enqueue time) or by the GPU?"""
import cProfile, os, pstats, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gridnext_amd as ga
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.rand((B, 3, 128, 128), generator=g, device=DEV)
y = torch.randint(0, 8, (B,), device=DEV)
f = ga.DenseNet(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64,
                bn_size=4, drop_rate=0).to(DEV).train()
opt = torch.optim.Adam(f.parameters(), lr=1e-3)
crit = nn.CrossEntropyLoss()
def step():
    opt.zero_grad()
    loss = crit(f(x), y)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("batch %d: enqueue %.2f ms/step, wall %.2f ms/step" % (B, t_enq / N * 1e3, t_all / N * 1e3))
# phases
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); te = time.perf_counter() - t; torch.cuda.synchronize()
    return r, te * 1e3, (time.perf_counter() - t) * 1e3
opt.zero_grad()
out, e1, w1 = timed(lambda: f(x))
loss = crit(out, y)
_, e2, w2 = timed(lambda: loss.backward())
_, e3, w3 = timed(lambda: opt.step())
print("forward enqueue %.2f wall %.2f | backward enqueue %.2f wall %.2f | adam enqueue %.2f wall %.2f" % (e1, w1, e2, w2, e3, w3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(14)
